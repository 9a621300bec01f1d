#define CIAO_T double
#include "mrhs_launch.inc"
