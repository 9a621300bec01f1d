#define CIAO_T float
#define CIAO_WS_ALG ciao::CA_SAGA
#include "chain_ws_launch.inc"
