#define CIAO_T float
#define CIAO_DMA_PART 4
#include "chain_dma_launch.inc"
