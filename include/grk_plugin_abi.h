/* include/grk_plugin_abi.h -- the Grok plugin boundary as seen from libgrokj2k_plugin.so.
 *
 * Grok (v8.0.2) dlopen()s "libgrokj2k_plugin.so" from the directory given to grk_initialize()
 * (src/lib/jp2/grok.cpp:549-576) and resolves the entry points below by name with dlsym
 * (grok.cpp:525-538, plugin/plugin_interface.h:46-143, plugin/minpf_plugin_manager.cpp:134-178).
 * A plugin normally compiles against grok.h; this repository must not carry reference sources,
 * so the part of that ABI the hot path touches is RESTATED here under our own type names
 * (gra_*).  Field order, types and array bounds are the ABI and therefore identical; the mirror
 * is verified field by field against the real header by tests/test_plugin_host.py::test_abi_mirror_compiled_against_grok_headers (sizeof /
 * offsetof of every struct through oracle/_ref, oracle/ref_harness/abi_check.cpp) -- if Grok's header changes, that test fails.
 *
 *   mirror type                      reference type (src/lib/jp2/grok.h)
 *   gra_plugin_pass                  grk_plugin_pass                 :1190-1194
 *   gra_plugin_code_block            grk_plugin_code_block           :1199-1212
 *   gra_plugin_precinct/band/...     grk_plugin_precinct ... tile    :1217-1263
 *   gra_plugin_init_info             grk_plugin_init_info            :1749-1752
 *   gra_progression                  grk_progression                 :386-431
 *   gra_raw_cparameters              grk_raw_cparameters             :440-447
 *   gra_cparameters                  grk_cparameters                 :451-573
 *   gra_encode_callback_info         plugin_encode_user_callback_info  plugin/plugin_interface.h:56-64
 *   gra_minpf_*                      minpf_* (plugin/minpf_plugin.h:25-60)
 */
#ifndef GRK_PLUGIN_ABI_H
#define GRK_PLUGIN_ABI_H
#include <stdint.h>
#include <stddef.h>
#include <stdbool.h>
#include "grok_amd.h"

#ifdef __cplusplus
extern "C" {
#endif

/* ---- constants of grok.h the structs are sized by ------------------------------------------ */
#define GRA_PATH_LEN                4096   /* GRK_PATH_LEN                 grok.h:100 */
#define GRA_J2K_MAXRLVLS            33     /* GRK_J2K_MAXRLVLS             grok.h:102 */
#define GRA_NUM_COMMENTS_SUPPORTED  256    /* GRK_NUM_COMMENTS_SUPPORTED   grok.h:351 */
#define GRA_CBLKSTY_HT              0x40   /* GRK_CBLKSTY_HT */
#define GRA_PLUGIN_STATE_DEBUG      0x1   /* grok.h:1738: the host compares every code-block with its own Tier-1 */
#define GRA_PLUGIN_STATE_NO_DEBUG   0x0    /* GRK_PLUGIN_STATE_NO_DEBUG    grok.h:1719 */
#define GRA_MAX_PASSES              67     /* grk_plugin_code_block.passes[] */

/* ---- data handed back to the host after encode (grok.h:1190-1263) -------------------------- */
typedef struct gra_plugin_pass {
    double distortionDecrease;
    size_t rate;
    size_t length;
} gra_plugin_pass;

typedef struct gra_plugin_code_block {
    uint32_t x0, y0, x1, y1;          /* debug info: band coordinates */
    unsigned int* contextStream;
    uint32_t numPix;
    uint8_t* compressedData;          /* owned by the plugin; host aliases it (plugin_bridge.cpp:198) */
    uint32_t compressedDataLength;
    size_t numBitPlanes;              /* 1 for HT: matches T1HT::compress (T1HT.cpp:123) */
    size_t numPasses;                 /* 1: cleanup pass only */
    gra_plugin_pass passes[GRA_MAX_PASSES];
    unsigned int sortedIndex;
} gra_plugin_code_block;

typedef struct gra_plugin_precinct {
    uint64_t numBlocks;
    gra_plugin_code_block** blocks;
} gra_plugin_precinct;

typedef struct gra_plugin_band {
    uint8_t orientation;
    uint64_t numPrecincts;
    gra_plugin_precinct** precincts;
    float stepsize;
} gra_plugin_band;

typedef struct gra_plugin_resolution {
    size_t level;
    size_t numBands;
    gra_plugin_band** band;
} gra_plugin_resolution;

typedef struct gra_plugin_tile_component {
    size_t numResolutions;
    gra_plugin_resolution** resolutions;
} gra_plugin_tile_component;

typedef struct gra_plugin_tile {
    uint32_t decompress_flags;
    size_t numComponents;
    gra_plugin_tile_component** tileComponents;
} gra_plugin_tile;

typedef struct gra_plugin_init_info {
    int32_t deviceId;
    bool verbose;
} gra_plugin_init_info;

/* ---- compress parameters (grok.h:386-573); enums are int-sized -------------------------------- */
typedef struct gra_progression {
    uint16_t layS, layE; uint8_t resS, resE; uint16_t compS, compE; uint64_t precS, precE;
    int32_t specifiedCompressionPocProg, progression;
    char progressionString[5];
    uint32_t tileno, tx0, tx1, ty0, ty1;
    uint16_t tpLayE; uint8_t tpResS, tpResE; uint16_t tpCompS, tpCompE; uint64_t tpPrecE;
    uint32_t tp_txS, tp_txE, tp_tyS, tp_tyE, dx, dy;
    uint16_t lay_temp; uint8_t res_temp; uint16_t comp_temp; uint64_t prec_temp;
    uint32_t tx0_temp, ty0_temp;
} gra_progression;

typedef struct gra_raw_comp_cparameters { uint32_t dx, dy; } gra_raw_comp_cparameters;
typedef struct gra_raw_cparameters {
    uint32_t width, height; uint16_t numcomps; uint8_t prec; bool sgnd;
    gra_raw_comp_cparameters* comps;
} gra_raw_cparameters;

typedef struct gra_cparameters {
    bool tile_size_on;
    uint32_t tx0, ty0, t_width, t_height;
    bool cp_disto_alloc, cp_fixed_quality;
    char* cp_comment[GRA_NUM_COMMENTS_SUPPORTED];
    uint16_t cp_comment_len[GRA_NUM_COMMENTS_SUPPORTED];
    bool cp_is_binary_comment[GRA_NUM_COMMENTS_SUPPORTED];
    size_t cp_num_comments;
    uint8_t csty;
    int32_t prog_order;
    gra_progression progression[GRA_J2K_MAXRLVLS];
    uint32_t numpocs;
    uint16_t tcp_numlayers;
    double tcp_rates[100];
    double tcp_distoratio[100];
    uint8_t numresolution;
    uint32_t cblockw_init, cblockh_init;
    uint8_t cblk_sty;
    bool isHT;
    bool irreversible;
    int32_t roi_compno;
    uint32_t roi_shift;
    uint32_t res_spec;
    uint32_t prcw_init[GRA_J2K_MAXRLVLS];
    uint32_t prch_init[GRA_J2K_MAXRLVLS];
    char infile[GRA_PATH_LEN];
    char outfile[GRA_PATH_LEN];
    uint32_t image_offset_x0, image_offset_y0, subsampling_dx, subsampling_dy;
    int32_t decod_format, cod_format;
    gra_raw_cparameters raw_cp;
    uint32_t max_comp_size;
    uint8_t tp_on, tp_flag, tcp_mct;
    void* mct_data;
    uint64_t max_cs_size;
    uint16_t rsiz, framerate;
    bool write_capture_resolution_from_file;
    double capture_resolution_from_file[2];
    bool write_capture_resolution;
    double capture_resolution[2];
    bool write_display_resolution;
    double display_resolution[2];
    uint32_t rateControlAlgorithm, numThreads;
    int32_t deviceId;
    uint32_t duration, kernelBuildOptions, repeats;
    bool writePLT, writeTLM, verbose;
} gra_cparameters;

/* what the plugin passes to the host's encode callback (plugin/plugin_interface.h:56-67) */
typedef struct gra_encode_callback_info {
    const char* input_file_name;
    bool outputFileNameIsRelative;
    const char* output_file_name;
    gra_cparameters* compressor_parameters;
    void* image;                       /* grk_image*; NULL = host loads the image itself */
    gra_plugin_tile* tile;
    int32_t error_code;
} gra_encode_callback_info;
typedef void (*gra_encode_callback)(gra_encode_callback_info* info);
typedef int32_t (*gra_decode_callback)(void* info /* PluginDecodeCallbackInfo*, C++ only */);

/* ---- decode side: what the host hands the plugin's init_decompressors_func (grok.h:1814-1815) ---------------
 *   gra_header_info   grk_header_info  grok.h:634-687   (leading fields spelled out, the rest opaque; same size)
 *   gra_image_comp    grk_image_comp   grok.h:866-891
 *   gra_image         grk_image        grok.h:907-929
 * PluginDecodeCallbackInfo itself (plugin/plugin_interface.h:86-130) has std::string members: it is mirrored in
 * C++ inside plugin.cpp, and its layout is checked by oracle/ref_harness/abi_check.cpp like everything here. */
#define GRA_DECODE_HEADER        (1u << 0)      /* grok.h:1249-1254 */
#define GRA_DECODE_T2            (1u << 1)
#define GRA_DECODE_T1            (1u << 2)
#define GRA_DECODE_POST_T1       (1u << 3)
#define GRA_PLUGIN_DECODE_CLEAN  (1u << 4)
#define GRA_HEADER_INFO_SIZE     11360
typedef struct gra_header_info {
    uint32_t cblockw_init, cblockh_init;
    bool     irreversible;
    uint32_t mct;
    uint16_t rsiz;
    uint32_t numresolutions;
    uint8_t  csty, cblk_sty;
    uint32_t prcw_init[GRA_J2K_MAXRLVLS], prch_init[GRA_J2K_MAXRLVLS];
    uint32_t tx0, ty0, t_width, t_height, t_grid_width, t_grid_height;
    uint16_t tcp_numlayers;
    uint64_t opaque[(GRA_HEADER_INFO_SIZE - 320) / 8];   /* xml, comments, asoc boxes: not touched */
} gra_header_info;
typedef struct gra_image_comp {
    void*    obj;                 /* grk_object */
    uint32_t dx, dy, w, stride, h, x0, y0;
    uint16_t Xcrg, Ycrg;
    uint8_t  prec;
    bool     sgnd;
    int32_t* data;
    int32_t  type, association;   /* GRK_COMPONENT_TYPE, GRK_COMPONENT_ASSOC */
} gra_image_comp;
typedef struct gra_image {
    void*    obj;
    uint32_t x0, y0, x1, y1;
    uint16_t numcomps;
    int32_t  color_space;         /* GRK_COLOR_SPACE */
    bool     color_applied, has_capture_resolution;
    double   capture_resolution[2];
    bool     has_display_resolution;
    double   display_resolution[2];
    void*    meta;
    gra_image_comp* comps;
} gra_image;
typedef int (*gra_init_decompressors_func)(gra_header_info* header_info, gra_image* image);
/* grk_image_cmptparm (grok.h:934-955): what grk_image_new() takes -- the plugin's self-check mode asks the host library
 * for the image it hands back with its coefficients */
typedef struct gra_image_cmptparm {
    uint32_t dx, dy, w, stride, h, x0, y0;
    uint8_t  prec;
    bool     sgnd;
} gra_image_cmptparm;
/* head of grk_decompress_parameters (grok.h:692-732 grk_dparameters, :754-760): the input path is all the plugin
 * reads out of it -- plugin_decompress takes the stream's QCD and the file size from the file (see plugin.cpp) */
typedef struct gra_dparameters {
    uint8_t  cp_reduce;
    uint16_t cp_layer;
    char     infile[GRA_PATH_LEN], outfile[GRA_PATH_LEN];
    int32_t  decod_format, cod_format;      /* GRK_SUPPORTED_FILE_FMT */
    uint32_t DA_x0, DA_x1, DA_y0, DA_y1;
    bool     m_verbose;
    uint16_t tile_index;
    uint32_t nb_tile_to_decompress, flags;
    int32_t  tileCacheStrategy;             /* GRK_TILE_CACHE_STRATEGY */
} gra_dparameters;
typedef struct gra_decompress_parameters_head {
    gra_dparameters core;
    char infile[GRA_PATH_LEN], outfile[GRA_PATH_LEN];
    int32_t decod_format, cod_format;       /* GRK_SUPPORTED_FILE_FMT (grok.h:59-72): input / output file type */
} gra_decompress_parameters_head;

/* ---- minimal plugin framework registration (plugin/minpf_plugin.h:25-60) --------------------- */
typedef struct gra_minpf_api_version { int32_t major, minor; } gra_minpf_api_version;
typedef struct gra_minpf_object_params { const char* id; const struct gra_minpf_platform_services* platformServices; } gra_minpf_object_params;
typedef void* (*gra_minpf_create_func)(gra_minpf_object_params*);
typedef int32_t (*gra_minpf_destroy_func)(void*);
typedef struct gra_minpf_register_params {
    gra_minpf_api_version version;
    gra_minpf_create_func createFunc;
    gra_minpf_destroy_func destroyFunc;
} gra_minpf_register_params;
typedef struct gra_minpf_platform_services {
    gra_minpf_api_version version;
    int32_t (*registerObject)(const char* nodeType, const gra_minpf_register_params* params);
    int32_t (*invokeService)(const char* serviceName, void* serviceParams);
} gra_minpf_platform_services;
typedef int32_t (*gra_minpf_exit_func)(void);

/* ================= entry points exported by libgrokj2k_plugin.so =================================
 * (the symbol list of the in-tree stub, src/lib/jp2_plugin/Plugin.cpp:19-125) */
gra_minpf_exit_func minpf_post_load_plugin(const char* pluginPath, const gra_minpf_platform_services* services);
bool     plugin_init(gra_plugin_init_info info);
int32_t  plugin_encode(gra_cparameters* params, gra_encode_callback callback);
int32_t  plugin_batch_encode(const char* input_dir, const char* output_dir, gra_cparameters* params, gra_encode_callback callback);
bool     plugin_is_batch_complete(void);
void     plugin_stop_batch_encode(void);
int32_t  plugin_decompress(void* decompress_parameters, gra_decode_callback callback);
int32_t  plugin_init_batch_decompress(const char* input_dir, const char* output_dir, void* decompress_parameters, gra_decode_callback callback);
int32_t  plugin_batch_decompress(void);
void     plugin_stop_batch_decompress(void);
uint32_t plugin_get_debug_state(void);
void     plugin_debug_mqc_next_cxd(void* mqc, uint32_t d);    /* name looked up by plugin_bridge.cpp:288 */
void     plugin_debug_next_cxd(void* mqc, uint32_t d);        /* name exported by the stub */
void     plugin_debug_mqc_next_plane(void* mqc);

/* how the last decode batch went: files decoded on the GPU / handed back to the host's own decoder / failed */
void grk_amd_plugin_batch_decode_counts(int32_t* gpu, int32_t* cpu, int32_t* failed);

/* ---- library-level drop-in (no file I/O): build the tile tree for grk_compress_with_plugin() ---
 * (grok.cpp:438).  `pixels`: one tile, layout of grk_amd_encode_tiles.
 * The tree, and the coded bytes its blocks point at, stay valid until ..._tile_destroy. */
gra_plugin_tile* grk_amd_plugin_tile_create(grk_amd_ctx* ctx, const grk_amd_tile_params* p,
                                            const void* pixels, int pixels_on_device);
void grk_amd_plugin_tile_destroy(gra_plugin_tile* tile);
/* The same for an image whose components are sub-sampled each in its own way (grk_image_comp::dx / dy, e.g. 4:2:0): `p` is the tile
 * on the REFERENCE grid, component c covers [ceil(x0 / dx_c), ceil(x1 / dx_c)) x [ceil(y0 / dy_c), ceil(y1 / dy_c)) of its own
 * samples (tile/TileProcessor.cpp:605-612); `planes` (host) holds the components back to back, each tight at its own size.  The tree
 * carries every component's own resolutions, precincts and blocks, as the host's tile does. */
gra_plugin_tile* grk_amd_plugin_tile_create_subsampled(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const uint8_t* comp_dx,
                                                       const uint8_t* comp_dy, const void* planes);
/* The rate-control hook: passes[0].distortionDecrease of every block of a tree just made by ..._tile_create, from
 * grk_amd_block_distortion (what compress_synch_with_plugin copies into the host's passes when the job has rate or quality targets,
 * plugin/plugin_bridge.cpp:214-226).  plugin_encode calls it for jobs with several layers or -r / -q targets. */
int grk_amd_plugin_tile_fill_distortion(grk_amd_ctx* ctx, gra_plugin_tile* tile);
/* device contexts the batch mode spreads files over (plugin_init: the device Grok named + the node's other GPUs, or what the
 * environment variable GRK_AMD_PLUGIN_DEVICES lists, e.g. "0,0": two contexts on GPU 0) */
uint32_t grk_amd_plugin_num_devices(void);

/* ---- the decode counterpart: a tile tree as the HOST fills it after its Tier-2 parse in
 * decompress_synch_plugin_with_host (plugin/plugin_bridge.cpp:63-76: per block compressedData,
 * compressedDataLength, numBitPlanes = block numbps, numPasses) is decoded on the GPU into `pixels`
 * (layout of grk_amd_decode_tiles).  p->reserved[0] = 1 for Part-1 (EBCOT) blocks, 0 for HT.
 * Returns 0, or a negative GRK_AMD_ERR_* so that the host keeps its CPU decoder. */
int grk_amd_plugin_tile_decode(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile,
                               void* pixels, int pixels_on_device);
/* The same for an HT stream whose QCD is not the one this library's encoder writes (another encoder's guard bits or
 * exponents): band_numbps[band] = expn_b + guard bits - 1 in QCD order (codestream/Quantizer.cpp:49-51), nbands =
 * 3 * levels + 1 -- the HT decoder's missing_msbs is band numbps - block numbps (T1DecompressScheduler.cpp:59) and
 * the host hands over block numbps only.  NULL / 0: the exponents of grk_amd_tile_layout (Grok's own HT streams). */
int grk_amd_plugin_tile_decode_qcd(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const gra_plugin_tile* tile,
                                   const uint8_t* band_numbps, uint32_t nbands, void* pixels, int pixels_on_device);

/* ... and for a tree whose components are sub-sampled each in its own way (the decode counterpart of
 * grk_amd_plugin_tile_create_subsampled): `p` = the tile on the reference grid, `planes` (host) receives the components back to back,
 * each tight at its own size. */
int grk_amd_plugin_tile_decode_subsampled(grk_amd_ctx* ctx, const grk_amd_tile_params* p, const uint8_t* comp_dx, const uint8_t* comp_dy,
                                          const gra_plugin_tile* tile, const uint8_t* band_numbps, uint32_t nbands, void* planes);

#ifdef __cplusplus
}
#endif
#endif /* GRK_PLUGIN_ABI_H */
