#define CIAO_T double
#include "chain_launch.inc"
