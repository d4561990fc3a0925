// oracle/ref_harness/ref_harness.cpp -- TEST INFRASTRUCTURE ONLY.
//
// Thin extern "C" shim over the REAL reference (Grok 8.0.2 compiled by oracle/Makefile into
// oracle/_ref/libgrokj2k_ref.so).  It lets the Python tests / bench.py's cpu_baseline leg call
//   * the reference's stage kernels directly (mct::compress_rev/irrev, dwt53/dwt97 row+column
//     kernels in the level loop of WaveletFwd.cpp:491-602, ojph_encode_codeblock, ojph_decode_codeblock)
//   * the reference's whole encoder/decoder through its public API, with the call sequence of
//     tests/test_tile_encoder.cpp:68-253 (+ HT flags), writing to a memory stream
//   * grk_compress_with_plugin(codec, tile) with a grk_plugin_tile tree built by OUR plugin
//     (the drop-in boundary).
// Nothing here is product code and nothing in grok_amd/ may link it.
#include <cstdint>
#include <cstring>
#include <cstdlib>
#include <chrono>
#include <atomic>
#include <vector>
#include <unistd.h>
#include <algorithm>
#include <cstdio>

// ojph headers first: grk_includes.h poisons malloc/free (util/MemManager.h:73)
#include "ojph_block_encoder.h"
#include "ojph_block_decoder.h"
#include "ojph_mem.h"
#include "grk_includes.h"

using namespace grk;

static bool g_inited = false;
static int g_verbose = 0;

static void msg_quiet(const char*, void*) {}
// warnings are counted: in the plugin's self-check mode (GRK_PLUGIN_STATE_DEBUG) every disagreement between the host's own
// Tier-1 and the plugin's code-blocks is reported as one (plugin_bridge.cpp:149-251)
static int g_warnings = 0;
static char g_last_warning[256] = {0};
static void msg_warn(const char* m, void*) { ++g_warnings; strncpy(g_last_warning, m ? m : "", sizeof(g_last_warning) - 1); if (g_verbose) fprintf(stderr, "[grok warning] %s\n", m); }
static void msg_err(const char* m, void*) { if (g_verbose) fprintf(stderr, "[grok] %s\n", m); }

extern "C" {

int ref_init(int threads, int verbose)
{
	g_verbose = verbose;
	if (!g_inited) {
		grk_initialize(nullptr, (uint32_t)threads);
		grk_set_info_handler(msg_quiet, nullptr);
		grk_set_warning_handler(msg_warn, nullptr);
		grk_set_error_handler(msg_err, nullptr);
		g_inited = true;
	}
	return (int)ThreadPool::get()->num_threads();
}

// ---------------------------------------------------------------- stage kernels
// the reference kernels use aligned SIMD loads: stage the caller's arrays in grkAlignedMalloc memory
static void mct_staged(int32_t* c0, int32_t* c1, int32_t* c2, uint64_t n, bool irrev)
{
	int32_t* p[3]; int32_t* u[3] = {c0, c1, c2};
	for (int i = 0; i < 3; ++i) { p[i] = (int32_t*)grkAlignedMalloc((n + 64) * 4); memcpy(p[i], u[i], n * 4); }
	if (irrev) mct::compress_irrev(p[0], p[1], p[2], n); else mct::compress_rev(p[0], p[1], p[2], n);
	for (int i = 0; i < 3; ++i) { memcpy(u[i], p[i], n * 4); grkAlignedFree(p[i]); }
}
void ref_rct(int32_t* c0, int32_t* c1, int32_t* c2, uint64_t n) { mct_staged(c0, c1, c2, n, false); }
void ref_ict(int32_t* c0, int32_t* c1, int32_t* c2, uint64_t n) { mct_staged(c0, c1, c2, n, true); }

} // extern "C"
template <typename T, typename DWT>
static void fwd_levels(T* user, uint32_t w, uint32_t h, uint32_t ustride, uint32_t levels, uint32_t x0 = 0, uint32_t y0 = 0)
{
	// aligned staging plane with the reference's own stride rule (util/MemManager.cpp:38-43)
	uint32_t stride = (w + 31u) & ~31u;
	T* plane = (T*)grkAlignedMalloc(((size_t)stride * h + 64) * sizeof(T));
	for (uint32_t y = 0; y < h; ++y) memcpy(plane + (size_t)y * stride, user + (size_t)y * ustride, w * sizeof(T));
	// resolution windows of a tile component at (x0, y0): [ceil(x0 / 2^l), ceil((x0 + w) / 2^l)); the parity of the
	// window's origin is what WaveletFwdImpl::encode_procedure passes as `even` (WaveletFwd.cpp:478-604)
	std::vector<uint32_t> rw(levels + 1), rh(levels + 1), ox(levels + 1), oy(levels + 1);
	for (uint32_t l = 0; l <= levels; ++l) {
		ox[l] = (uint32_t)(((uint64_t)x0 + (1ull << l) - 1) >> l);
		oy[l] = (uint32_t)(((uint64_t)y0 + (1ull << l) - 1) >> l);
		rw[l] = (uint32_t)(((uint64_t)x0 + w + (1ull << l) - 1) >> l) - ox[l];
		rh[l] = (uint32_t)(((uint64_t)y0 + h + (1ull << l) - 1) >> l) - oy[l];
	}
	size_t tmpn = (size_t)std::max(w, h) * 8 + 64;
	T* tmp = (T*)grkAlignedMalloc(tmpn * sizeof(T));
	DWT dwt;
	for (uint32_t l = 0; l < levels; ++l) {
		uint32_t cw = rw[l], ch = rh[l];
		const bool even_x = (ox[l] & 1u) == 0, even_y = (oy[l] & 1u) == 0;
		// vertical pass, 8 columns at a time (WaveletFwd.cpp:491-507)
		uint32_t j = 0;
		for (; j + 8 - 1 < cw; j += 8)
			dwt.encode_and_deinterleave_v(plane + j, tmp, ch, even_y, stride, 8);
		if (j < cw)
			dwt.encode_and_deinterleave_v(plane + j, tmp, ch, even_y, stride, cw - j);
		// horizontal pass (WaveletFwd.cpp:553-561)
		for (uint32_t r = 0; r < ch; ++r)
			dwt.encode_and_deinterleave_h_one_row(plane + (size_t)r * stride, tmp, cw, even_x);
	}
	grkAlignedFree(tmp);
	for (uint32_t y = 0; y < h; ++y) memcpy(user + (size_t)y * ustride, plane + (size_t)y * stride, w * sizeof(T));
	grkAlignedFree(plane);
}

extern "C" {
void ref_dwt53_fwd(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels)
{ fwd_levels<int32_t, dwt53>(plane, w, h, stride, levels); }
void ref_dwt97_fwd(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels)
{ fwd_levels<float, dwt97>(plane, w, h, stride, levels); }
void ref_dwt53_fwd_at(int32_t* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0)
{ fwd_levels<int32_t, dwt53>(plane, w, h, stride, levels, x0, y0); }
void ref_dwt97_fwd_at(float* plane, uint32_t w, uint32_t h, uint32_t stride, uint32_t levels, uint32_t x0, uint32_t y0)
{ fwd_levels<float, dwt97>(plane, w, h, stride, levels, x0, y0); }

void ref_dwt53_row(int32_t* row, uint32_t n, int even)
{
	int32_t* a = (int32_t*)grkAlignedMalloc((n + 16) * 4); memcpy(a, row, n * 4);
	int32_t* tmp = (int32_t*)grkAlignedMalloc((n + 16) * 4);
	dwt53 d; d.encode_and_deinterleave_h_one_row(a, tmp, n, even != 0);
	memcpy(row, a, n * 4); grkAlignedFree(a); grkAlignedFree(tmp);
}
void ref_dwt97_row(float* row, uint32_t n, int even)
{
	float* a = (float*)grkAlignedMalloc((n + 16) * 4); memcpy(a, row, n * 4);
	float* tmp = (float*)grkAlignedMalloc((n + 16) * sizeof(float));
	dwt97 d; d.encode_and_deinterleave_h_one_row(a, tmp, n, even != 0);
	memcpy(row, a, n * 4); grkAlignedFree(a); grkAlignedFree(tmp);
}

// HT cleanup encode of one block. `sm` = sign-magnitude words as produced by T1HT::preCompress.
int32_t ref_ht_encode_block(uint32_t* sm, uint32_t kmax, uint32_t w, uint32_t h, uint32_t stride,
							uint8_t* out, uint32_t cap)
{
	static thread_local ojph::mem_elastic_allocator* elastic = nullptr;
	if (elastic) { delete elastic; elastic = nullptr; }
	elastic = new ojph::mem_elastic_allocator(1048576);
	ojph::coded_lists* coded = nullptr;
	uint32_t lengths[2] = {0, 0};
	ojph::local::ojph_encode_codeblock(sm, kmax, 1, w, h, stride, lengths, elastic, coded);
	if (lengths[0] > cap) return -1;
	memcpy(out, coded->buf, lengths[0]);
	return (int32_t)lengths[0];
}

// HT cleanup decode; `missing_msbs` as Grok passes it (Kmax-1, T1DecompressScheduler.cpp:59).
int32_t ref_ht_decode_block(const uint8_t* coded, uint32_t len, uint32_t missing_msbs,
							uint32_t w, uint32_t h, uint32_t* out)
{
	// the decoder works on whole quad pairs: give it a padded scratch plane and copy the block out
	std::vector<uint8_t> buf(len + 32, 0);
	memcpy(buf.data() + 8, coded, len);
	const uint32_t stride = ((w + 7u) & ~7u) + 8u;
	std::vector<uint32_t> tmp((size_t)stride * (h + 4), 0);
	bool ok = ojph::local::ojph_decode_codeblock(buf.data() + 8, tmp.data(), missing_msbs, 1, len, 0, w, h, stride);
	for (uint32_t y = 0; y < h; ++y) memcpy(out + (size_t)y * w, tmp.data() + (size_t)y * stride, w * 4);
	return ok ? 0 : -1;
}

// The same with the refinement passes: coded = cleanup segment (lengths1 bytes) followed by the SigProp / MagRef segment
// (lengths2 bytes); num_passes 1..3.  Grok never calls the decoder this way (T1HT.cpp:158-166: lengths2 = 0); the
// function itself handles it (ojph_block_decoder.cpp:1054-1058, :1627-2100), which is what pins oracle/ht_refine_oracle.c.
int32_t ref_ht_decode_block_passes(const uint8_t* coded, uint32_t lengths1, uint32_t lengths2, uint32_t num_passes,
								   uint32_t missing_msbs, uint32_t w, uint32_t h, uint32_t* out)
{
	const uint32_t len = lengths1 + lengths2;
	std::vector<uint8_t> buf(len + 64, 0);
	memcpy(buf.data() + 16, coded, len);
	const uint32_t stride = ((w + 7u) & ~7u) + 8u;
	std::vector<uint32_t> tmp((size_t)stride * (h + 8), 0);
	bool ok = ojph::local::ojph_decode_codeblock(buf.data() + 16, tmp.data(), missing_msbs, num_passes, lengths1, lengths2, w, h, stride);
	for (uint32_t y = 0; y < h; ++y) memcpy(out + (size_t)y * w, tmp.data() + (size_t)y * stride, w * 4);
	return ok ? 0 : -1;
}

// ---------------------------------------------------------------- whole codec
struct EncCfg {
	int32_t C, W, H, TW, TH, prec, irrev, numres, ht, mode; // mode 0: grk_compress_tile, 1: grk_compress(image data)
	int32_t rate_algo;                                        // parameters.rateControlAlgorithm
	int32_t cblk_w, cblk_h;                                   // 0 = default 64
	int32_t cblk_sty;                                         // Part-1 code-block style bits (grk_compress -M)
};

static void fill_params(grk_cparameters& p, const EncCfg& c)
{
	grk_compress_set_default_params(&p);
	p.tile_size_on = true;
	p.tx0 = 0; p.ty0 = 0;
	p.t_width = (uint32_t)c.TW; p.t_height = (uint32_t)c.TH;
	p.irreversible = c.irrev != 0;
	p.numresolution = (uint32_t)c.numres;
	p.prog_order = GRK_LRCP;
	p.tcp_mct = (c.C >= 3) ? 1 : 0;
	if (const char* e = getenv("REF_TCP_MCT")) { if (atoi(e) != 255) p.tcp_mct = (uint8_t)atoi(e); }
	if (c.ht) { p.isHT = true; p.cblk_sty = GRK_CBLKSTY_HT; }
	else if (c.cblk_sty) p.cblk_sty = (uint8_t)c.cblk_sty;
	p.rateControlAlgorithm = (uint32_t)c.rate_algo;
	if (c.cblk_w) p.cblockw_init = (uint32_t)c.cblk_w;
	if (c.cblk_h) p.cblockh_init = (uint32_t)c.cblk_h;
	// grk_compress -X / -L: pointer marker segments (TLM in the main header, PLT in the tile-part headers)
	if (const char* e = getenv("REF_WRITE_TLM")) p.writeTLM = atoi(e) != 0;
	if (const char* e = getenv("REF_WRITE_PLT")) p.writePLT = atoi(e) != 0;
	// grk_compress -p / -S / -E: progression order, SOP and EPH markers
	if (const char* e = getenv("REF_PROG_ORDER")) p.prog_order = (GRK_PROG_ORDER)atoi(e);
	if (const char* e = getenv("REF_CSTY")) p.csty = (uint8_t)(p.csty | atoi(e));
	// grk_compress -c [w,h],[w,h],...: precinct sizes from the highest resolution down (REF_PRECINCTS="w,h,w,h,...")
	if (const char* e = getenv("REF_PRECINCTS")) {
		uint32_t n = 0; const char* q = e;
		while (*q && n < GRK_J2K_MAXRLVLS) {
			unsigned pw = 0, ph = 0; int used = 0;
			if (sscanf(q, "%u,%u%n", &pw, &ph, &used) != 2) break;
			p.prcw_init[n] = pw; p.prch_init[n] = ph; ++n;
			q += used; if (*q == ',') ++q;
		}
		if (n) { p.res_spec = n; p.csty |= 0x01; }
	}
	// grk_compress -r r1,r2,...: quality layers by compression ratio, largest first; a last ratio of 1 = everything that is left
	// (grk_compress.cpp: "-r 20,10,1"; cp_disto_alloc, tcp_rates[], tcp_numlayers -- the last layer's rate 0 = lossless)
	if (const char* e = getenv("REF_LAYERS")) {
		uint32_t n = 0; const char* q = e;
		while (*q && n < 100) {
			double r = 0; int used = 0;
			if (sscanf(q, "%lf%n", &r, &used) != 1) break;
			p.tcp_rates[n++] = r <= 1.0 ? 0.0 : r;
			q += used; if (*q == ',') ++q;
		}
		if (n) { p.tcp_numlayers = (uint16_t)n; p.cp_disto_alloc = true; }
	}
	// grk_compress -s dx,dy: the sub-sampling factors of the components on the reference grid (all components alike)
	if (const char* e = getenv("REF_SUBSAMPLING")) {
		unsigned dx = 1, dy = 1;
		if (sscanf(e, "%u,%u", &dx, &dy) == 2 && dx >= 1 && dy >= 1) { p.subsampling_dx = dx; p.subsampling_dy = dy; }
	}
	// grk_compress -d: the image area's origin on the canonical grid (the tile grid stays anchored at 0, 0)
	if (const char* e = getenv("REF_IMG_X0")) p.image_offset_x0 = (uint32_t)atoi(e);
	if (const char* e = getenv("REF_IMG_Y0")) p.image_offset_y0 = (uint32_t)atoi(e);
}

static grk_image* make_image(const EncCfg& c, bool alloc)
{
	std::vector<grk_image_cmptparm> cp((size_t)c.C);
	memset(cp.data(), 0, sizeof(grk_image_cmptparm) * cp.size());
	const uint32_t ix0 = getenv("REF_IMG_X0") ? (uint32_t)atoi(getenv("REF_IMG_X0")) : 0;
	const uint32_t iy0 = getenv("REF_IMG_Y0") ? (uint32_t)atoi(getenv("REF_IMG_Y0")) : 0;
	// (sub-sampled components, grk_compress -s: the image area on the reference grid is what the image readers make of a w x h
	//  component, x1 = x0 + (w - 1) dx + 1, src/bin/image_format: every component then has ceil(x1 / dx) - ceil(x0 / dx) = w columns)
	unsigned sdx = 1, sdy = 1;
	if (const char* e = getenv("REF_SUBSAMPLING")) { if (sscanf(e, "%u,%u", &sdx, &sdy) != 2 || !sdx || !sdy) sdx = sdy = 1; }
	for (auto& q : cp) {
		q.dx = sdx; q.dy = sdy; q.w = (uint32_t)c.W; q.h = (uint32_t)c.H;
		q.x0 = ix0; q.y0 = iy0; q.prec = (uint8_t)c.prec; q.sgnd = false;
	}
	// components sub-sampled each in its own way (REF_COMP_SUBSAMPLING="dx0,dy0,dx1,dy1,...": a raw / yuv image, grk_compress -F
	// w,h,c,prec,u@1x1:2x2:2x2, image_format/RAWFormat.cpp:255-290): W x H is the image area on the reference grid, component c has
	// ceil(x1 / dx) - ceil(x0 / dx) columns -- what grk_image_new computes from x0 / x1 is overwritten by cmptparm w / h, so both are set
	bool per_comp = false;
	if (const char* e = getenv("REF_COMP_SUBSAMPLING")) {
		const char* q = e;
		for (auto& cpi : cp) {
			unsigned dx = 1, dy = 1; int used = 0;
			if (sscanf(q, "%u,%u%n", &dx, &dy, &used) != 2 || !dx || !dy) break;
			cpi.dx = dx; cpi.dy = dy;
			cpi.w = (ix0 + (uint32_t)c.W + dx - 1) / dx - (ix0 + dx - 1) / dx;
			cpi.h = (iy0 + (uint32_t)c.H + dy - 1) / dy - (iy0 + dy - 1) / dy;
			q += used; if (*q == ',') ++q;
			per_comp = true;
		}
	}
	auto img = grk_image_new((uint16_t)c.C, cp.data(), c.C >= 3 ? GRK_CLRSPC_SRGB : GRK_CLRSPC_GRAY, alloc);
	if (!img) return nullptr;
	img->x0 = ix0; img->y0 = iy0; img->x1 = ix0 + ((uint32_t)c.W - 1) * sdx + 1; img->y1 = iy0 + ((uint32_t)c.H - 1) * sdy + 1;
	if (per_comp) { img->x1 = ix0 + (uint32_t)c.W; img->y1 = iy0 + (uint32_t)c.H; }
	return img;
}

// Encode `pixels` (component-major planar, tightly packed, ceil(prec/8) bytes per sample, whole
// image) with the reference CPU encoder. Returns coded length or <0. `secs` = wall time of the
// compress calls only (steady_clock), the figure BASELINE.md §3 asks for.
int64_t ref_encode(const EncCfg* cfg, const uint8_t* pixels, uint8_t* out, uint64_t cap, double* secs,
				   void* plugin_tile /* grk_plugin_tile* or null */)
{
	const EncCfg& c = *cfg;
	grk_cparameters param;
	fill_params(param, c);
	const int bps = (c.prec + 7) / 8;
	bool use_image_data = (c.mode == 1) || plugin_tile;
	grk_image* image = make_image(c, true);   // multi-tile paths dereference comp->data (TileProcessor.cpp:1137-1147)
	if (!image) return -2;
	if (use_image_data) {
		// (components back to back, each with its own size: all W x H unless REF_COMP_SUBSAMPLING says otherwise)
		const uint8_t* src = pixels;
		for (int k = 0; k < c.C; ++k) {
			auto comp = image->comps + k;
			const int cw = (int)comp->w, chh = (int)comp->h;
			for (int y = 0; y < chh; ++y)
				for (int x = 0; x < cw; ++x) {
					size_t i = (size_t)y * cw + x;
					comp->data[(size_t)y * comp->stride + x] =
						bps == 1 ? (int32_t)src[i] : (int32_t)((const uint16_t*)src)[i];
				}
			src += (size_t)cw * chh * bps;
		}
	}
	grk_stream* stream = grk_stream_create_mem_stream(out, cap, false, false);
	grk_codec* codec = grk_compress_create(GRK_CODEC_J2K, stream);
	int64_t rc = -3;
	double t = 0;
	do {
		if (!codec) break;
		if (!grk_compress_init(codec, &param, image)) { rc = -4; break; }
		auto t0 = std::chrono::steady_clock::now();
		if (!grk_compress_start(codec)) { rc = -5; break; }
		if (plugin_tile) {
			if (!grk_compress_with_plugin(codec, (grk_plugin_tile*)plugin_tile)) { rc = -6; break; }
		} else if (c.mode == 1) {
			if (!grk_compress(codec)) { rc = -6; break; }
		} else {
			int tcols = (c.W + c.TW - 1) / c.TW, trows = (c.H + c.TH - 1) / c.TH;
			std::vector<uint8_t> tilebuf;
			bool ok = true;
			for (int ty = 0; ty < trows && ok; ++ty)
				for (int tx = 0; tx < tcols && ok; ++tx) {
					int x0 = tx * c.TW, y0 = ty * c.TH;
					int tw = std::min(c.TW, c.W - x0), th = std::min(c.TH, c.H - y0);
					const uint8_t* src = pixels;
					uint64_t tsize = (uint64_t)tw * th * c.C * bps;
					if (tcols * trows > 1) {
						tilebuf.resize(tsize);
						for (int k = 0; k < c.C; ++k)
							for (int y = 0; y < th; ++y)
								memcpy(tilebuf.data() + ((size_t)k * th + y) * tw * bps,
									   pixels + (((size_t)k * c.H + y0 + y) * c.W + x0) * bps, (size_t)tw * bps);
						src = tilebuf.data();
					}
					ok = grk_compress_tile(codec, (uint16_t)(ty * tcols + tx), (uint8_t*)src, tsize);
				}
			if (!ok) { rc = -6; break; }
		}
		if (!grk_compress_end(codec)) { rc = -7; break; }
		t = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
		rc = (int64_t)grk_stream_get_write_mem_stream_length(stream);
	} while (0);
	if (secs) *secs = t;
	grk_object_unref(stream);
	grk_object_unref(codec);
	grk_object_unref(&image->obj);
	return rc;
}

// Decode a whole codestream with the reference decoder into int32 planes (C x H x W, tight).
int32_t ref_decode(const uint8_t* j2k, uint64_t len, int32_t* out, int32_t C, int32_t W, int32_t H)
{
	grk_dparameters dp;
	grk_decompress_set_default_params(&dp);
	if (const char* e = getenv("REF_MAX_LAYERS")) dp.cp_layer = (uint16_t)atoi(e);      // grk_decompress -l: the first quality layers only
	grk_stream* stream = grk_stream_create_mem_stream((uint8_t*)j2k, len, false, true);
	grk_codec* codec = grk_decompress_create(GRK_CODEC_J2K, stream);
	int32_t rc = -1;
	do {
		if (!codec) break;
		if (!grk_decompress_init(codec, &dp)) { rc = -2; break; }
		grk_header_info hi; memset(&hi, 0, sizeof(hi));
		if (!grk_decompress_read_header(codec, &hi)) { rc = -3; break; }
		if (!grk_decompress(codec, nullptr)) { rc = -4; break; }
		grk_image* img = grk_decompress_get_composited_image(codec);
		if (!img || img->numcomps != C) { rc = -5; break; }
		if (getenv("REF_COMP_SUBSAMPLING")) {       // components of their own sizes, back to back in `out` (room for C x H x W)
			int32_t* dst = out;
			for (int k = 0; k < C; ++k) {
				auto comp = img->comps + k;
				if ((int)comp->w > W || (int)comp->h > H || !comp->data) { rc = -6; break; }
				for (uint32_t y = 0; y < comp->h; ++y, dst += comp->w)
					memcpy(dst, comp->data + (size_t)y * comp->stride, (size_t)comp->w * 4);
			}
			if (rc == -6) break;
			grk_decompress_end(codec);
			rc = 0;
			break;
		}
		for (int k = 0; k < C; ++k) {
			auto comp = img->comps + k;
			if ((int)comp->w != W || (int)comp->h != H || !comp->data) { rc = -6; break; }
			for (int y = 0; y < H; ++y)
				memcpy(out + ((size_t)k * H + y) * W, comp->data + (size_t)y * comp->stride, (size_t)W * 4);
		}
		if (rc == -6) break;
		grk_decompress_end(codec);
		rc = 0;
	} while (0);
	grk_object_unref(stream);
	grk_object_unref(codec);
	return rc;
}

// Windowed decode (grk_decompress_set_window, grok.h; WaveletReverse.cpp:1466-2213 does the partial synthesis):
// out = C planes of (x1 - x0) x (y1 - y0) samples.
int32_t ref_decode_window(const uint8_t* j2k, uint64_t len, int32_t* out, int32_t C, uint32_t x0, uint32_t y0, uint32_t x1, uint32_t y1)
{
	grk_dparameters dp;
	grk_decompress_set_default_params(&dp);
	grk_stream* stream = grk_stream_create_mem_stream((uint8_t*)j2k, len, false, true);
	grk_codec* codec = grk_decompress_create(GRK_CODEC_J2K, stream);
	int32_t rc = -1;
	const int W = (int)(x1 - x0), H = (int)(y1 - y0);
	do {
		if (!codec) break;
		if (!grk_decompress_init(codec, &dp)) { rc = -2; break; }
		grk_header_info hi; memset(&hi, 0, sizeof(hi));
		if (!grk_decompress_read_header(codec, &hi)) { rc = -3; break; }
		if (!grk_decompress_set_window(codec, x0, y0, x1, y1)) { rc = -7; break; }
		if (!grk_decompress(codec, nullptr)) { rc = -4; break; }
		grk_image* img = grk_decompress_get_composited_image(codec);
		if (!img || img->numcomps != C) { rc = -5; break; }
		for (int k = 0; k < C; ++k) {
			auto comp = img->comps + k;
			if ((int)comp->w != W || (int)comp->h != H || !comp->data) { rc = -6; break; }
			for (int y = 0; y < H; ++y)
				memcpy(out + ((size_t)k * H + y) * W, comp->data + (size_t)y * comp->stride, (size_t)W * 4);
		}
		if (rc == -6) break;
		grk_decompress_end(codec);
		rc = 0;
	} while (0);
	grk_object_unref(stream);
	grk_object_unref(codec);
	return rc;
}

// ---- the file-oriented plugin protocol: grk_plugin_compress(params, callback) (grok.cpp:628) ----
// The callback below is what src/bin/jp2/grk_compress.cpp:1604-1987 does in essence: build the
// image, create codec + stream, grk_compress_init/start, grk_compress_with_plugin(codec, tile), end.
static const EncCfg* g_cb_cfg = nullptr;
static const uint8_t* g_cb_pixels = nullptr;
static uint8_t* g_cb_out = nullptr;
static uint64_t g_cb_cap = 0;
static int64_t g_cb_len = -100;

static bool host_compress_callback(grk_plugin_compress_user_callback_info* info)
{
	const EncCfg& c = *g_cb_cfg;
	g_cb_len = -101;
	if (!info || !info->tile) { if (info) info->error_code = 1; return false; }
	const int bps = (c.prec + 7) / 8;
	// as grk_compress.cpp:1609-1636: an image the plugin passes is used as it is (its self-check mode hands over its
	// sub-band coefficients that way), otherwise the host loads the source
	const bool plugin_image = info->image != nullptr;
	grk_image* image = plugin_image ? info->image : make_image(c, true);
	if (plugin_image && getenv("REF_DEBUG_PERTURB")) image->comps[0].data[5 * image->comps[0].stride + 7] ^= 1;   // (test: must be noticed)
	for (int k = 0; k < c.C && !plugin_image; ++k) {
		auto comp = image->comps + k;
		const uint8_t* src = g_cb_pixels + (size_t)k * c.W * c.H * bps;
		for (int y = 0; y < c.H; ++y)
			for (int x = 0; x < c.W; ++x) {
				size_t i = (size_t)y * c.W + x;
				comp->data[(size_t)y * comp->stride + x] = bps == 1 ? (int32_t)src[i] : (int32_t)((const uint16_t*)src)[i];
			}
	}
	// "Decide if MCT should be used" (grk_compress.cpp:1817-1820): 255 = not set on the command line
	if (info->compressor_parameters->tcp_mct == 255) info->compressor_parameters->tcp_mct = (image->numcomps >= 3) ? 1 : 0;
	grk_stream* stream = grk_stream_create_mem_stream(g_cb_out, g_cb_cap, false, false);
	grk_codec* codec = grk_compress_create(GRK_CODEC_J2K, stream);
	bool ok = codec && grk_compress_init(codec, info->compressor_parameters, image) && grk_compress_start(codec) &&
			  grk_compress_with_plugin(codec, info->tile) && grk_compress_end(codec);
	g_cb_len = ok ? (int64_t)grk_stream_get_write_mem_stream_length(stream) : -102;
	info->error_code = ok ? 0 : 1;
	grk_object_unref(stream);
	grk_object_unref(codec);
	if (!plugin_image) grk_object_unref(&image->obj);
	return ok;
}

// returns coded length (>=0), or the plugin's refusal code (<0)
int64_t ref_plugin_compress_file(const EncCfg* cfg, const uint8_t* pixels, const char* infile, uint8_t* out, uint64_t cap)
{
	grk_cparameters param;
	fill_params(param, *cfg);
	strncpy(param.infile, infile, GRK_PATH_LEN - 1);
	// (single-tile images: the plugin calls back and the callback writes into `out`; images of several tiles: the plugin writes
	//  the codestream file itself, next to the input)
	snprintf(param.outfile, GRK_PATH_LEN, "%s.j2k", infile);
	// grk_compress leaves tcp_mct at 255 ("not set") until its callback has loaded the image (grk_compress.cpp:1836,
	// :1708-1720): REF_TCP_MCT=255 hands the plugin that sentinel, the callback below resolves it as the CLI does
	if (const char* e = getenv("REF_TCP_MCT")) param.tcp_mct = (uint8_t)atoi(e);
	g_cb_cfg = cfg; g_cb_pixels = pixels; g_cb_out = out; g_cb_cap = cap; g_cb_len = -100;
	remove(param.outfile);
	int32_t rc = grk_plugin_compress(&param, host_compress_callback);
	if (rc != 0) return rc < 0 ? rc : -rc;
	if (g_cb_len == -100) {              // no callback: the plugin wrote the file
		FILE* f = fopen(param.outfile, "rb");
		if (!f) return -200;
		const size_t n = fread(out, 1, cap, f);
		fclose(f);
		return (int64_t)n;
	}
	return g_cb_len;
}

// ---- batch compress: grk_plugin_batch_compress(in_dir, out_dir, params, callback) (grok.cpp:683-707) -- what
// `grk_compress -y in_dir -a out_dir` does (grk_compress.cpp:1540-1590): the plugin walks the directory and calls back once
// per image; the callback loads the image named by info->input_file_name itself and writes info->output_file_name.
static bool read_pnm_planar(const char* path, std::vector<int32_t>& planes, int& C, int& W, int& H, int& prec)
{
	FILE* f = fopen(path, "rb");
	if (!f) return false;
	char magic[3] = {0, 0, 0};
	unsigned w = 0, h = 0, maxv = 0;
	bool ok = fscanf(f, "%2s %u %u %u", magic, &w, &h, &maxv) == 4 && fgetc(f) != EOF && (!strcmp(magic, "P5") || !strcmp(magic, "P6"));
	if (ok) {
		C = magic[1] == '6' ? 3 : 1; W = (int)w; H = (int)h;
		prec = 1; while ((1u << prec) <= maxv) ++prec;
		const size_t bps = prec > 8 ? 2 : 1, n = (size_t)w * h;
		std::vector<uint8_t> raw(n * C * bps);
		ok = fread(raw.data(), 1, raw.size(), f) == raw.size();
		planes.resize(n * C);
		for (int c = 0; c < C && ok; ++c)
			for (size_t i = 0; i < n; ++i)
				planes[c * n + i] = bps == 1 ? raw[i * C + c] : (raw[(i * C + c) * 2] << 8 | raw[(i * C + c) * 2 + 1]);
	}
	fclose(f);
	return ok;
}
static std::atomic<int> g_batch_ok{0}, g_batch_bad{0};
static bool host_batch_compress_callback(grk_plugin_compress_user_callback_info* info)
{
	if (!info || !info->tile || !info->input_file_name || !info->output_file_name) { g_batch_bad++; if (info) info->error_code = 1; return false; }
	std::vector<int32_t> planes;
	EncCfg c{};
	int prec = 0;
	if (!read_pnm_planar(info->input_file_name, planes, c.C, c.W, c.H, prec)) { g_batch_bad++; info->error_code = 1; return false; }
	c.prec = prec;
	grk_image* image = make_image(c, true);
	for (int k = 0; k < c.C; ++k)
		for (int y = 0; y < c.H; ++y)
			memcpy(image->comps[k].data + (size_t)y * image->comps[k].stride, planes.data() + ((size_t)k * c.H + y) * c.W, (size_t)c.W * 4);
	grk_cparameters param = *info->compressor_parameters;           // (every image of the batch starts from the same parameters)
	if (param.tcp_mct == 255) param.tcp_mct = (image->numcomps >= 3) ? 1 : 0;
	std::vector<uint8_t> out((size_t)c.W * c.H * c.C * 4 + (1u << 20));
	grk_stream* stream = grk_stream_create_mem_stream(out.data(), out.size(), false, false);
	grk_codec* codec = grk_compress_create(GRK_CODEC_J2K, stream);
	bool ok = codec && grk_compress_init(codec, &param, image) && grk_compress_start(codec) &&
			  grk_compress_with_plugin(codec, info->tile) && grk_compress_end(codec);
	if (ok) {
		FILE* f = fopen(info->output_file_name, "wb");
		ok = f && fwrite(out.data(), 1, grk_stream_get_write_mem_stream_length(stream), f) == grk_stream_get_write_mem_stream_length(stream);
		if (f) fclose(f);
	}
	info->error_code = ok ? 0 : 1;
	(ok ? g_batch_ok : g_batch_bad)++;
	grk_object_unref(stream);
	grk_object_unref(codec);
	grk_object_unref(&image->obj);
	return ok;
}
// returns the number of files written (callbacks that succeeded), or < 0: refused / timed out / a callback failed
int32_t ref_plugin_batch_compress(const EncCfg* cfg, const char* in_dir, const char* out_dir, int timeout_s)
{
	static grk_cparameters param;        // (read by the plugin's worker after this call returns)
	fill_params(param, *cfg);
	param.tile_size_on = false;          // the images of a directory differ in size
	param.tcp_mct = 255;
	g_batch_ok = 0; g_batch_bad = 0;
	int32_t rc = grk_plugin_batch_compress(in_dir, out_dir, &param, host_batch_compress_callback);
	if (rc != 0) return rc < 0 ? rc : -rc;
	for (int i = 0; i < timeout_s * 10 && !grk_plugin_is_batch_complete(); ++i) usleep(100000);
	const bool done = grk_plugin_is_batch_complete();
	grk_plugin_stop_batch_compress();
	if (!done) return -1000;
	return g_batch_bad.load() ? -2000 - g_batch_bad.load() : g_batch_ok.load();
}

// ---- the decode protocol: grk_plugin_decompress(params, callback) (grok.cpp:727-743) ------------------------
// The callback below is what src/bin/jp2/grk_decompress.cpp:971-1008 (decompress_callback -> preProcess /
// postProcess) does in essence, with a memory stream instead of a file and a buffer instead of an image writer.
static const uint8_t* g_dcb_j2k = nullptr;
static uint64_t g_dcb_len = 0;
static int32_t* g_dcb_out = nullptr;
static int32_t g_dcb_C = 0, g_dcb_W = 0, g_dcb_H = 0;
static int g_dcb_stage[4] = {0, 0, 0, 0};     // header, t2, post-t1, clean calls seen

static int32_t host_decompress_callback(grk_plugin_decompress_callback_info* info)
{
	if (!info) return -1;
	if (info->decompress_flags & GRK_PLUGIN_DECODE_CLEAN) {
		g_dcb_stage[3]++;
		if (info->stream) grk_object_unref(info->stream);
		info->stream = nullptr;
		if (info->codec) grk_object_unref(info->codec);
		info->codec = nullptr;
		info->image = nullptr;
		return 0;
	}
	if (info->decompress_flags & GRK_DECODE_HEADER) {
		g_dcb_stage[0]++;
		// batch mode: the plugin names the file (grk_decompress.cpp:1018-1019); otherwise the test's memory buffer
		if (!info->stream)
			info->stream = (!g_dcb_j2k && info->input_file_name) ? grk_stream_create_file_stream(info->input_file_name, 1 << 20, true)
																 : grk_stream_create_mem_stream((uint8_t*)g_dcb_j2k, g_dcb_len, false, true);
		if (!info->stream) return 1;
		if (!info->codec) {
			info->codec = grk_decompress_create(GRK_CODEC_J2K, info->stream);
			if (!info->codec) return 1;
			if (!grk_decompress_init(info->codec, &info->decompressor_parameters->core)) return 1;
		}
		if (!grk_decompress_read_header(info->codec, &info->header_info)) return 1;
		info->image = grk_decompress_get_composited_image(info->codec);
		if (info->init_decompressors_func) { int rc = info->init_decompressors_func(&info->header_info, info->image); if (rc || !(info->decompress_flags & (GRK_DECODE_T2 | GRK_DECODE_T1))) return rc; }
		else if (!(info->decompress_flags & (GRK_DECODE_T2 | GRK_DECODE_T1))) return 0;
	}
	if (info->decompress_flags & (GRK_DECODE_T2 | GRK_DECODE_T1)) {
		g_dcb_stage[1]++;
		if (!info->codec) return 1;
		if (!info->tile && !(info->decompress_flags & GRK_DECODE_T1)) return 1;      // (no tile: the host decodes everything itself)
		if (info->tile) info->tile->decompress_flags = info->decompress_flags;
		if (getenv("REF_HARNESS_DEBUG")) grk_set_error_handler([](const char* m, void*) { fprintf(stderr, "[grk error] %s\n", m); }, nullptr);
		if (!grk_decompress_set_window(info->codec, 0, 0, 0, 0)) { if (getenv("REF_HARNESS_DEBUG")) fprintf(stderr, "set_window failed\n"); return 1; }
		if (!grk_decompress(info->codec, info->tile)) { if (getenv("REF_HARNESS_DEBUG")) fprintf(stderr, "grk_decompress failed\n"); return 1; }
		if (!grk_decompress_end(info->codec)) { if (getenv("REF_HARNESS_DEBUG")) fprintf(stderr, "decompress_end failed\n"); return 1; }
		// T2 | POST_T1 is the plugin's Tier-2 stage (it asks for both because of defect D11): the image is stored in the
		// POST_T1 call that follows its own decode; only a full host decode (T1 set) goes on to store the image here
		if (!(info->decompress_flags & GRK_DECODE_T1) || !(info->decompress_flags & GRK_DECODE_POST_T1)) return 0;
	}
	if (info->decompress_flags & GRK_DECODE_POST_T1) {
		g_dcb_stage[2]++;
		auto img = info->image;
		if (!img) return 1;
		if (!g_dcb_out) {
			// batch mode: the decoded image goes to the file the plugin names, as "C W H\n" + int32 planes (test format)
			if (!info->output_file_name) return 1;
			FILE* f = fopen(info->output_file_name, "wb");
			if (!f) return 1;
			fprintf(f, "%d %d %d\n", (int)img->numcomps, (int)img->comps[0].w, (int)img->comps[0].h);
			for (uint16_t k = 0; k < img->numcomps; ++k)
				for (uint32_t y = 0; y < img->comps[k].h; ++y)
					fwrite(img->comps[k].data + (size_t)y * img->comps[k].stride, 4, img->comps[k].w, f);
			fclose(f);
			++g_dcb_stage[2];
			return 0;
		}
		if (img->numcomps != g_dcb_C) return 1;
		if (getenv("REF_COMP_SUBSAMPLING")) {      // components of their own sizes, back to back (room for C x H x W)
			int32_t* dst = g_dcb_out;
			for (int k = 0; k < g_dcb_C; ++k) {
				auto comp = img->comps + k;
				if ((int)comp->w > g_dcb_W || (int)comp->h > g_dcb_H || !comp->data) return 1;
				for (uint32_t y = 0; y < comp->h; ++y, dst += comp->w) memcpy(dst, comp->data + (size_t)y * comp->stride, (size_t)comp->w * 4);
			}
			return 0;
		}
		for (int k = 0; k < g_dcb_C; ++k) {
			auto comp = img->comps + k;
			if ((int)comp->w != g_dcb_W || (int)comp->h != g_dcb_H || !comp->data) return 1;
			for (int y = 0; y < g_dcb_H; ++y)
				memcpy(g_dcb_out + ((size_t)k * g_dcb_H + y) * g_dcb_W, comp->data + (size_t)y * comp->stride, (size_t)g_dcb_W * 4);
		}
		return 0;
	}
	return -1;
}

// returns the plugin's answer (0 = decoded by the plugin), stages[] = how often each protocol stage was called
// infile: where the same bytes lie as a file (grk_decompress -i sets parameters->infile, grk_decompress.cpp:552; a plugin
// may read the main header from it), or null
int32_t ref_plugin_decompress(const uint8_t* j2k, uint64_t len, int32_t* out, int32_t C, int32_t W, int32_t H, int32_t* stages,
							  const char* infile)
{
	grk_decompress_parameters param;
	memset(&param, 0, sizeof(param));
	grk_decompress_set_default_params(&param.core);
	param.decod_format = GRK_J2K_FMT;
	if (infile) strncpy(param.infile, infile, GRK_PATH_LEN - 1);
	g_dcb_j2k = j2k; g_dcb_len = len; g_dcb_out = out; g_dcb_C = C; g_dcb_W = W; g_dcb_H = H;
	memset(g_dcb_stage, 0, sizeof(g_dcb_stage));
	int32_t rc = grk_plugin_decompress(&param, host_decompress_callback);
	if (stages) memcpy(stages, g_dcb_stage, sizeof(g_dcb_stage));
	return rc;
}

// grk_plugin_init_batch_decompress + grk_plugin_batch_decompress over a directory, polled as grk_decompress.cpp:874-900 does;
// returns 0 when the batch ran to completion, the plugin's refusal otherwise
int32_t ref_plugin_batch_decompress(const char* in_dir, const char* out_dir, int timeout_s)
{
	static grk_decompress_parameters param;
	memset(&param, 0, sizeof(param));
	grk_decompress_set_default_params(&param.core);
	param.decod_format = GRK_J2K_FMT;
	param.cod_format = GRK_RAW_FMT;
	g_dcb_j2k = nullptr; g_dcb_len = 0; g_dcb_out = nullptr;
	memset(g_dcb_stage, 0, sizeof(g_dcb_stage));
	int32_t rc = grk_plugin_init_batch_decompress(in_dir, out_dir, &param, host_decompress_callback);
	if (rc) return rc;
	rc = grk_plugin_batch_decompress();
	if (rc) return rc;
	for (int i = 0; i < timeout_s * 10 && !grk_plugin_is_batch_complete(); ++i) usleep(100000);
	const bool done = grk_plugin_is_batch_complete();
	grk_plugin_stop_batch_decompress();
	return done ? 0 : -7;
}

// Load a real plugin .so through the reference's own minpf loader (grk_initialize(pluginPath)),
// then report what the host sees. Used by the boundary test.
int ref_plugin_load(const char* dir, int threads)
{
	bool ok = grk_initialize(dir, (uint32_t)threads);
	g_inited = true;
	grk_set_info_handler(msg_quiet, nullptr);
	grk_set_warning_handler(msg_warn, nullptr);
	grk_set_error_handler(msg_err, nullptr);
	return ok ? 1 : 0;
}
int ref_plugin_init(int device, int verbose)
{
	grk_plugin_init_info info; info.deviceId = device; info.verbose = verbose != 0;
	return grk_plugin_init(info) ? 1 : 0;
}
uint32_t ref_plugin_debug_state(void) { return grk_plugin_get_debug_state(); }
int ref_warning_count(int reset, char* last, int cap)
{
	const int n = g_warnings;
	if (last && cap > 0) { strncpy(last, g_last_warning, (size_t)cap - 1); last[cap - 1] = 0; }
	if (reset) { g_warnings = 0; g_last_warning[0] = 0; }
	return n;
}

// sizes/offsets of the ABI structs, for the mirror check in tests/test_abi.py
uint64_t ref_abi_sizeof(int which)
{
	switch (which) {
	case 0: return sizeof(grk_plugin_pass);
	case 1: return sizeof(grk_plugin_code_block);
	case 2: return sizeof(grk_plugin_precinct);
	case 3: return sizeof(grk_plugin_band);
	case 4: return sizeof(grk_plugin_resolution);
	case 5: return sizeof(grk_plugin_tile_component);
	case 6: return sizeof(grk_plugin_tile);
	case 7: return sizeof(grk_cparameters);
	case 8: return offsetof(grk_plugin_code_block, compressedData);
	case 9: return offsetof(grk_plugin_code_block, passes);
	case 10: return offsetof(grk_plugin_band, stepsize);
	case 11: return sizeof(grk_plugin_init_info);
	case 12: return sizeof(grk_plugin_compress_user_callback_info);
	case 13: return offsetof(grk_cparameters, isHT);
	case 14: return offsetof(grk_cparameters, tcp_mct);
	case 15: return offsetof(grk_cparameters, numresolution);
	case 16: return offsetof(grk_cparameters, irreversible);
	case 17: return offsetof(grk_cparameters, infile);
	case 18: return offsetof(grk_cparameters, outfile);
	case 19: return offsetof(grk_cparameters, t_width);
	case 20: return offsetof(grk_cparameters, cblockw_init);
	case 21: return offsetof(grk_cparameters, deviceId);
	case 22: return offsetof(grk_cparameters, rateControlAlgorithm);
	case 23: return offsetof(grk_cparameters, cblk_sty);
	default: return 0;
	}
}

} // extern "C"

// ---------------------------------------------------------------- Part-1 (EBCOT/MQ) block coder
// Drives the reference's own T1 (t1/t1_part1/T1.cpp) on one code-block, the way T1Part1::compress /
// ::decompress do (t1/t1_part1/T1Part1.cpp:33-151): reversible path, cblk_sty = 0, no rate control.
#include "t1_common.h"
extern "C" {

// coef: w x h int32 (row stride `stride`).  Returns coded length; *numpasses / *numbps as the encoder
// reports them (numbps = magnitude bit-planes actually coded).
int32_t ref_t1_encode_block(const int32_t* coef, uint32_t w, uint32_t h, uint32_t stride, uint32_t orient,
							uint8_t* out, uint32_t cap, uint32_t* numpasses, uint32_t* numbps)
{
	grk::T1 t1(true, w, h);
	if (!t1.alloc(w, h)) return -1;
	auto d = t1.getUncompressedData();
	uint32_t maxv = 0;
	for (uint32_t y = 0; y < h; ++y)
		for (uint32_t x = 0; x < w; ++x) {
			int32_t temp = coef[(size_t)y * stride + x] * (1 << T1_NMSEDEC_FRACBITS);
			temp = (int32_t)to_smr(temp);
			maxv = std::max<uint32_t>(maxv, smr_abs(temp));
			d[(size_t)y * w + x] = temp;
		}
	std::vector<uint8_t> buf((size_t)w * h * 8 + 4096, 0);
	grk::cblk_enc c;
	memset(&c, 0, sizeof(c));
	c.x0 = 0; c.y0 = 0; c.x1 = w; c.y1 = h;
	c.data = buf.data() + 2;           // the coder writes one byte before the start (mqc_init_enc)
	t1.compress_cblk(&c, maxv, (uint8_t)orient, 0, 0, 1, 1.0, 0, nullptr, 0, false);
	uint32_t len = c.numPassesTotal ? c.passes[c.numPassesTotal - 1].rate : 0;
	*numpasses = c.numPassesTotal; *numbps = c.numbps;
	int32_t rc = -1;
	if (len <= cap) { memcpy(out, c.data, len); rc = (int32_t)len; }
	t1.code_block_enc_deallocate(&c);
	return rc;
}

// out: w x h int32 in the decoder's own representation (one extra fractional bit: value*2 +- 1)
int32_t ref_t1_decode_block(const uint8_t* coded, uint32_t len, uint32_t numpasses, uint32_t numbps,
							uint32_t orient, uint32_t w, uint32_t h, int32_t* out)
{
	grk::T1 t1(false, w, h);
	DecompressCodeblock cblk;
	cblk.setRect(grkRectU32(0, 0, w, h));
	if (!cblk.alloc()) return -1;
	cblk.numbps = numbps;
	auto seg = cblk.nextSegment();
	seg->numpasses = numpasses; seg->len = len; seg->maxpasses = numpasses;
	std::vector<int32_t> data((size_t)w * h, 0);
	t1.attachUncompressedData(data.data(), w, h);
	t1.allocCompressedData(len + 8);
	memcpy(t1.getCompressedDataBuffer(), coded, len);
	bool ok = t1.decompress_cblk(&cblk, t1.getCompressedDataBuffer(), (uint8_t)orient, 0);
	memcpy(out, data.data(), data.size() * 4);
	return ok ? 0 : -1;
}

// Part-1 encoder with a code-block style: also returns each pass's cumulative rate and termination flag, from which
// the codeword segments follow (a segment ends at every terminated pass and at the last pass).
int32_t ref_t1_encode_block_sty(const int32_t* coef, uint32_t w, uint32_t h, uint32_t stride, uint32_t orient, uint32_t cblksty,
								uint8_t* out, uint32_t cap, uint32_t* numpasses, uint32_t* numbps, uint32_t* pass_rate,
								uint8_t* pass_term, uint32_t pass_cap)
{
	grk::T1 t1(true, w, h);
	if (!t1.alloc(w, h)) return -1;
	auto d = t1.getUncompressedData();
	uint32_t maxv = 0;
	for (uint32_t y = 0; y < h; ++y)
		for (uint32_t x = 0; x < w; ++x) {
			int32_t temp = coef[(size_t)y * stride + x] * (1 << T1_NMSEDEC_FRACBITS);
			temp = (int32_t)to_smr(temp);
			maxv = std::max<uint32_t>(maxv, smr_abs(temp));
			d[(size_t)y * w + x] = temp;
		}
	std::vector<uint8_t> buf((size_t)w * h * 8 + 4096, 0);
	grk::cblk_enc c;
	memset(&c, 0, sizeof(c));
	c.x0 = 0; c.y0 = 0; c.x1 = w; c.y1 = h;
	c.data = buf.data() + 2;
	t1.compress_cblk(&c, maxv, (uint8_t)orient, 0, 0, 1, 1.0, cblksty, nullptr, 0, false);
	uint32_t len = c.numPassesTotal ? c.passes[c.numPassesTotal - 1].rate : 0;
	*numpasses = c.numPassesTotal; *numbps = c.numbps;
	int32_t rc = -1;
	if (len <= cap && c.numPassesTotal <= pass_cap) {
		memcpy(out, c.data, len); rc = (int32_t)len;
		for (uint32_t i = 0; i < c.numPassesTotal; ++i) { pass_rate[i] = c.passes[i].rate; pass_term[i] = c.passes[i].term ? 1 : 0; }
	}
	t1.code_block_enc_deallocate(&c);
	return rc;
}

int32_t ref_t1_decode_block_sty(const uint8_t* coded, uint32_t nsegs, const uint32_t* seg_len, const uint32_t* seg_passes,
								uint32_t numbps, uint32_t orient, uint32_t cblksty, uint32_t w, uint32_t h, int32_t* out)
{
	grk::T1 t1(false, w, h);
	DecompressCodeblock cblk;
	cblk.setRect(grkRectU32(0, 0, w, h));
	if (!cblk.alloc()) return -1;
	cblk.numbps = numbps;
	uint32_t len = 0;
	for (uint32_t i = 0; i < nsegs; ++i) {
		auto seg = cblk.nextSegment();
		seg->numpasses = seg_passes[i]; seg->len = seg_len[i]; seg->maxpasses = seg_passes[i];
		len += seg_len[i];
	}
	std::vector<int32_t> data((size_t)w * h, 0);
	t1.attachUncompressedData(data.data(), w, h);
	t1.allocCompressedData(len + 8);
	memcpy(t1.getCompressedDataBuffer(), coded, len);
	bool ok = t1.decompress_cblk(&cblk, t1.getCompressedDataBuffer(), (uint8_t)orient, cblksty);
	memcpy(out, data.data(), data.size() * 4);
	return ok ? 0 : -1;
}

} // extern "C"
