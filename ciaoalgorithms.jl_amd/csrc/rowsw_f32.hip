#define CIAO_T float
#include "rowsw_launch.inc"
