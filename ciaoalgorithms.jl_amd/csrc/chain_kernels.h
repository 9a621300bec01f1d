// chain_kernels.h -- everything the sequential inner loops need (SURVEY.md section 8a rows S3, G3, and F3/F4 with small batches),
// by family since round 5: chain_common.h says why the chains are ONE persistent workgroup with its state in registers.
#pragma once

#include "chain_common.h"
#include "chain_reg_kernels.h"
#include "chain_dma_kernels.h"
#include "afinito_kernels.h"
#include "vector_kernels.h"
