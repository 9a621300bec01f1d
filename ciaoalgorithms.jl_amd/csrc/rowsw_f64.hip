#define CIAO_T double
#include "rowsw_launch.inc"
