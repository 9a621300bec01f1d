#define CIAO_T float
#include "chain_cdma_launch.inc"
