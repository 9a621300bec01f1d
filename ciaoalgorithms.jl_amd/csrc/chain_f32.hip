#define CIAO_T float
#include "chain_launch.inc"
