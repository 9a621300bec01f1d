#define CIAO_T float
#define CIAO_DMA_PART 3
#include "chain_dma_launch.inc"
