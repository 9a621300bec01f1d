#define CIAO_T double
#define CIAO_DMA_PART 0
#include "chain_dma_launch.inc"
