#define CIAO_T float
#include "mrhs_launch.inc"
