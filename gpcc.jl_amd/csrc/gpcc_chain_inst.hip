// gpcc_chain_inst.hip -- the persistent few-evaluation kernel as its own translation unit (it compiles beside the tile kernels
// instead of behind them).
#include "gpcc_chain.hip.h"

hipError_t gpcc_chain_configure()
{
    return hipFuncSetAttribute((const void *)gpcc_chain_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, GPCC_CHAIN_LDS_BYTES);
}

void gpcc_chain_launch(const GpccCtx &c, const GpccGroup &g, const GpccChainArgs &a, unsigned grid, hipStream_t s)
{
    gpcc_chain_kernel<<<grid, GPCC_CHAIN_THREADS, GPCC_CHAIN_LDS_BYTES, s>>>(c, g, a);
}
