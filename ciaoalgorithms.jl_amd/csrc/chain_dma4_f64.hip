#define CIAO_T double
#define CIAO_DMA_PART 4
#include "chain_dma_launch.inc"
