#define CIAO_T float
#include "rowsm_launch.inc"
