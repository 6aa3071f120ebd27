// placeholder until the LDS-resident chain kernel lands (next commit)
#include <hip/hip_runtime.h>
#include "../../include/zpaq_hip.h"
#include "zpq_common.h"
extern "C" int zpq_chain_blocks_per_wg(const DModel *) { return 0; }
extern "C" int zpq_chain_max_wgs(const DModel *, int) { return 0; }
extern "C" int zpq_launch_chain(const DBatch *, const DModel *, int, int, hipStream_t) { return ZPQ_E_INTERNAL; }
extern "C" const char *zpq_chain_kernel_name(const DModel *, int) { return ""; }
