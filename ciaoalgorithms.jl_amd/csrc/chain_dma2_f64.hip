#define CIAO_T double
#define CIAO_DMA_PART 2
#include "chain_dma_launch.inc"
