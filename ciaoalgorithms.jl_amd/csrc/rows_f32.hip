#define CIAO_T float
#include "rows_launch.inc"
