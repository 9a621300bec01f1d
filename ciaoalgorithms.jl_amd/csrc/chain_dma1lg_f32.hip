#define CIAO_T float
#define CIAO_DMA_PART 1
#define CIAO_DMA_LOSS 1
#include "chain_dma_launch.inc"
