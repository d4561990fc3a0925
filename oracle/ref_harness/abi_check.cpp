// oracle/ref_harness/abi_check.cpp -- TEST INFRASTRUCTURE ONLY.
// Compile-time proof that include/grk_plugin_abi.h (our restatement of the plugin ABI) has exactly
// the layout of the reference's own headers.  Built by `make ref` wherever /root/reference exists;
// a mismatch is a compile error.
#include <cstddef>
#include "grk_includes.h"
#include "plugin_interface.h"
#include "minpf_plugin.h"
#include "grk_plugin_abi.h"

#define SAME_SIZE(A, B) static_assert(sizeof(A) == sizeof(B), "sizeof " #A " != " #B)
#define SAME_OFF(A, B, F) static_assert(offsetof(A, F) == offsetof(B, F), "offsetof " #A "." #F)

SAME_SIZE(gra_plugin_pass, grk_plugin_pass);
SAME_OFF(gra_plugin_pass, grk_plugin_pass, distortionDecrease); SAME_OFF(gra_plugin_pass, grk_plugin_pass, rate);
SAME_OFF(gra_plugin_pass, grk_plugin_pass, length);

SAME_SIZE(gra_plugin_code_block, grk_plugin_code_block);
SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, x0); SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, y1);
SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, contextStream); SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, numPix);
SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, compressedData); SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, compressedDataLength);
SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, numBitPlanes); SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, numPasses);
SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, passes); SAME_OFF(gra_plugin_code_block, grk_plugin_code_block, sortedIndex);

SAME_SIZE(gra_plugin_precinct, grk_plugin_precinct);
SAME_OFF(gra_plugin_precinct, grk_plugin_precinct, numBlocks); SAME_OFF(gra_plugin_precinct, grk_plugin_precinct, blocks);
SAME_SIZE(gra_plugin_band, grk_plugin_band);
SAME_OFF(gra_plugin_band, grk_plugin_band, orientation); SAME_OFF(gra_plugin_band, grk_plugin_band, numPrecincts);
SAME_OFF(gra_plugin_band, grk_plugin_band, precincts); SAME_OFF(gra_plugin_band, grk_plugin_band, stepsize);
SAME_SIZE(gra_plugin_resolution, grk_plugin_resolution);
SAME_OFF(gra_plugin_resolution, grk_plugin_resolution, level); SAME_OFF(gra_plugin_resolution, grk_plugin_resolution, numBands);
SAME_OFF(gra_plugin_resolution, grk_plugin_resolution, band);
SAME_SIZE(gra_plugin_tile_component, grk_plugin_tile_component);
SAME_OFF(gra_plugin_tile_component, grk_plugin_tile_component, numResolutions);
SAME_OFF(gra_plugin_tile_component, grk_plugin_tile_component, resolutions);
SAME_SIZE(gra_plugin_tile, grk_plugin_tile);
SAME_OFF(gra_plugin_tile, grk_plugin_tile, decompress_flags); SAME_OFF(gra_plugin_tile, grk_plugin_tile, numComponents);
SAME_OFF(gra_plugin_tile, grk_plugin_tile, tileComponents);
SAME_SIZE(gra_plugin_init_info, grk_plugin_init_info);
SAME_OFF(gra_plugin_init_info, grk_plugin_init_info, deviceId); SAME_OFF(gra_plugin_init_info, grk_plugin_init_info, verbose);

SAME_SIZE(gra_progression, grk_progression);
SAME_OFF(gra_progression, grk_progression, precS); SAME_OFF(gra_progression, grk_progression, progression);
SAME_OFF(gra_progression, grk_progression, progressionString); SAME_OFF(gra_progression, grk_progression, tileno);
SAME_OFF(gra_progression, grk_progression, tpPrecE); SAME_OFF(gra_progression, grk_progression, dy);
SAME_OFF(gra_progression, grk_progression, prec_temp); SAME_OFF(gra_progression, grk_progression, ty0_temp);
SAME_SIZE(gra_raw_cparameters, grk_raw_cparameters);
SAME_OFF(gra_raw_cparameters, grk_raw_cparameters, sgnd); SAME_OFF(gra_raw_cparameters, grk_raw_cparameters, comps);

SAME_SIZE(gra_cparameters, grk_cparameters);
#define CP(F) SAME_OFF(gra_cparameters, grk_cparameters, F)
CP(tile_size_on); CP(tx0); CP(ty0); CP(t_width); CP(t_height); CP(cp_disto_alloc); CP(cp_fixed_quality);
CP(cp_comment); CP(cp_comment_len); CP(cp_is_binary_comment); CP(cp_num_comments); CP(csty); CP(prog_order);
CP(progression); CP(numpocs); CP(tcp_numlayers); CP(tcp_rates); CP(tcp_distoratio); CP(numresolution);
CP(cblockw_init); CP(cblockh_init); CP(cblk_sty); CP(isHT); CP(irreversible); CP(roi_compno); CP(roi_shift);
CP(res_spec); CP(prcw_init); CP(prch_init); CP(infile); CP(outfile); CP(image_offset_x0); CP(image_offset_y0);
CP(subsampling_dx); CP(subsampling_dy); CP(decod_format); CP(cod_format); CP(raw_cp); CP(max_comp_size);
CP(tp_on); CP(tp_flag); CP(tcp_mct); CP(mct_data); CP(max_cs_size); CP(rsiz); CP(framerate);
CP(write_capture_resolution_from_file); CP(capture_resolution_from_file); CP(write_capture_resolution);
CP(capture_resolution); CP(write_display_resolution); CP(display_resolution); CP(rateControlAlgorithm);
CP(numThreads); CP(deviceId); CP(duration); CP(kernelBuildOptions); CP(repeats); CP(writePLT); CP(writeTLM); CP(verbose);

SAME_SIZE(gra_encode_callback_info, grk::plugin_encode_user_callback_info);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, input_file_name);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, outputFileNameIsRelative);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, output_file_name);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, compressor_parameters);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, image);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, tile);
SAME_OFF(gra_encode_callback_info, grk::plugin_encode_user_callback_info, error_code);

SAME_SIZE(gra_minpf_register_params, grk::minpf_register_params);
SAME_OFF(gra_minpf_register_params, grk::minpf_register_params, createFunc);
SAME_OFF(gra_minpf_register_params, grk::minpf_register_params, destroyFunc);
SAME_SIZE(gra_minpf_platform_services, grk::minpf_platform_services);
SAME_OFF(gra_minpf_platform_services, grk::minpf_platform_services, registerObject);
SAME_OFF(gra_minpf_platform_services, grk::minpf_platform_services, invokeService);

static_assert(GRA_PATH_LEN == GRK_PATH_LEN && GRA_J2K_MAXRLVLS == GRK_J2K_MAXRLVLS &&
              GRA_NUM_COMMENTS_SUPPORTED == GRK_NUM_COMMENTS_SUPPORTED && GRA_CBLKSTY_HT == GRK_CBLKSTY_HT, "constants");

// decode side
SAME_SIZE(gra_header_info, grk_header_info);
static_assert(sizeof(grk_header_info) == GRA_HEADER_INFO_SIZE, "GRA_HEADER_INFO_SIZE");
SAME_OFF(gra_header_info, grk_header_info, cblockw_init); SAME_OFF(gra_header_info, grk_header_info, cblockh_init);
SAME_OFF(gra_header_info, grk_header_info, irreversible); SAME_OFF(gra_header_info, grk_header_info, mct);
SAME_OFF(gra_header_info, grk_header_info, rsiz); SAME_OFF(gra_header_info, grk_header_info, numresolutions);
SAME_OFF(gra_header_info, grk_header_info, csty); SAME_OFF(gra_header_info, grk_header_info, cblk_sty);
SAME_OFF(gra_header_info, grk_header_info, prcw_init); SAME_OFF(gra_header_info, grk_header_info, prch_init);
SAME_OFF(gra_header_info, grk_header_info, tx0); SAME_OFF(gra_header_info, grk_header_info, t_width);
SAME_OFF(gra_header_info, grk_header_info, t_grid_width); SAME_OFF(gra_header_info, grk_header_info, t_grid_height);
SAME_OFF(gra_header_info, grk_header_info, tcp_numlayers);
SAME_SIZE(gra_image_comp, grk_image_comp);
SAME_OFF(gra_image_comp, grk_image_comp, dx); SAME_OFF(gra_image_comp, grk_image_comp, w); SAME_OFF(gra_image_comp, grk_image_comp, stride);
SAME_OFF(gra_image_comp, grk_image_comp, h); SAME_OFF(gra_image_comp, grk_image_comp, x0); SAME_OFF(gra_image_comp, grk_image_comp, y0);
SAME_OFF(gra_image_comp, grk_image_comp, prec); SAME_OFF(gra_image_comp, grk_image_comp, sgnd); SAME_OFF(gra_image_comp, grk_image_comp, data);
SAME_SIZE(gra_image, grk_image);
SAME_OFF(gra_image, grk_image, x0); SAME_OFF(gra_image, grk_image, y1); SAME_OFF(gra_image, grk_image, numcomps);
SAME_OFF(gra_image, grk_image, comps);
static_assert(GRA_DECODE_HEADER == GRK_DECODE_HEADER && GRA_DECODE_T2 == GRK_DECODE_T2 && GRA_DECODE_T1 == GRK_DECODE_T1 &&
              GRA_DECODE_POST_T1 == GRK_DECODE_POST_T1 && GRA_PLUGIN_DECODE_CLEAN == GRK_PLUGIN_DECODE_CLEAN, "decode flags");

// layout of the reference's C++ decode callback record, for comparison with the plugin's own mirror
// (grk_amd_plugin_decode_info_layout in libgrokj2k_plugin.so; tests/test_plugin_host.py)
extern "C" size_t ref_decode_info_layout(int which)
{
    using I = grk::PluginDecodeCallbackInfo;
    switch (which) {
    case 0: return sizeof(I);
    case 1: return offsetof(I, init_decompressors_func);
    case 2: return offsetof(I, inputFile);
    case 3: return offsetof(I, outputFile);
    case 4: return offsetof(I, decod_format);
    case 5: return offsetof(I, stream);
    case 6: return offsetof(I, codec);
    case 7: return offsetof(I, decompressor_parameters);
    case 8: return offsetof(I, header_info);
    case 9: return offsetof(I, image);
    case 10: return offsetof(I, plugin_owns_image);
    case 11: return offsetof(I, tile);
    case 12: return offsetof(I, error_code);
    case 13: return offsetof(I, decompress_flags);
    case 14: return offsetof(I, user_data);
    default: return 0;
    }
}

extern "C" int ref_abi_mirror_checked(void) { return 1; }

// decode parameters: the plugin reads the input path only (plugin.cpp: decompress_file)
SAME_SIZE(gra_dparameters, grk_dparameters);
SAME_OFF(gra_dparameters, grk_dparameters, infile); SAME_OFF(gra_dparameters, grk_dparameters, outfile);
SAME_OFF(gra_dparameters, grk_dparameters, tileCacheStrategy);
static_assert(offsetof(gra_decompress_parameters_head, core) == offsetof(grk_decompress_parameters, core), "core");
static_assert(offsetof(gra_decompress_parameters_head, infile) == offsetof(grk_decompress_parameters, infile), "infile");
static_assert(offsetof(gra_decompress_parameters_head, outfile) == offsetof(grk_decompress_parameters, outfile), "outfile");

SAME_SIZE(gra_image_cmptparm, grk_image_cmptparm);
SAME_OFF(gra_image_cmptparm, grk_image_cmptparm, stride); SAME_OFF(gra_image_cmptparm, grk_image_cmptparm, prec);
SAME_OFF(gra_image_cmptparm, grk_image_cmptparm, sgnd);
static_assert(GRA_PLUGIN_STATE_DEBUG == GRK_PLUGIN_STATE_DEBUG, "debug state");
static_assert(offsetof(gra_decompress_parameters_head, cod_format) == offsetof(grk_decompress_parameters, cod_format), "cod_format");
