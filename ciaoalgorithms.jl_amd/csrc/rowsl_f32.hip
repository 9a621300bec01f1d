#define CIAO_T float
#include "rowsl_launch.inc"
