#define CIAO_T double
#include "rows_launch.inc"
