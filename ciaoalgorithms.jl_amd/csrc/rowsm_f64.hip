#define CIAO_T double
#include "rowsm_launch.inc"
