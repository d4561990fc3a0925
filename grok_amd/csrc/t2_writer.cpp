// placeholder, replaced below
#include "../../include/grok_amd.h"
extern "C" int64_t grk_amd_write_codestream(const grk_amd_tile_params*, uint32_t, uint32_t, const grk_amd_coded_block*, const uint8_t*, uint8_t*, uint64_t) { return GRK_AMD_ERR_UNSUPPORTED; }
