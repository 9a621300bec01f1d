#define CIAO_T double
#include "rowsl_launch.inc"
