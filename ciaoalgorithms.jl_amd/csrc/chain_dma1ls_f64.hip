#define CIAO_T double
#define CIAO_DMA_PART 1
#define CIAO_DMA_LOSS 0
#include "chain_dma_launch.inc"
