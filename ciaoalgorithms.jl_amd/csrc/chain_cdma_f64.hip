#define CIAO_T double
#include "chain_cdma_launch.inc"
