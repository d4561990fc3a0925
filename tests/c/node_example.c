/* The C host of INTEGRATION.md 5a, compiled against include/grok_amd.h and linked with libgrok_amd.so (tests/test_capi_host.py):
 * a small image of 2 x 2 tiles over the device list given on the command line (default: all GPUs of the node), both forms of
 * the exchange; prints "no device" and exits 3 where there is no GPU.  usage: node_example [dev ...] */
#include "grok_amd.h"
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

int main(int argc, char** argv)
{
    int devs[16]; uint32_t nd = 0;
    for (int i = 1; i < argc && nd < 16; ++i) devs[nd++] = atoi(argv[i]);
    grk_amd_node* node = NULL;
    int rc = grk_amd_node_create(nd ? devs : NULL, nd, 0, &node);
    if (rc == GRK_AMD_ERR_NO_DEVICE) { printf("no device\n"); return 3; }
    if (rc != GRK_AMD_OK) { printf("node_create failed: %d\n", rc); return 1; }
    const uint32_t W = 512, H = 384, T = 256;
    grk_amd_image_layout im = {0, 0, W, H, 0, 0, T, T};
    grk_amd_tile_params base;
    memset(&base, 0, sizeof base);
    base.num_comps = 3; base.prec = 8; base.mct = 1; base.num_levels = 4; base.cblk_w_exp = 6; base.cblk_h_exp = 6;
    const size_t npx = (size_t)3 * W * H;
    uint8_t* px = (uint8_t*)grk_amd_host_alloc(grk_amd_node_ctx(node, 0), npx);
    if (!px) { printf("host_alloc failed\n"); return 1; }
    uint32_t s = 12345;
    for (size_t i = 0; i < npx; ++i) { s = s * 1664525u + 1013904223u; px[i] = (uint8_t)((i % W + i / W) / 8 + ((s >> 24) & 7)); }
    const uint64_t cap = npx * 2 + (1u << 20);
    uint8_t* a = (uint8_t*)malloc(cap); uint8_t* b = (uint8_t*)malloc(cap);
    const int64_t na = grk_amd_node_encode_image(node, &im, &base, px, GRK_AMD_CS_TLM, a, cap);
    const int64_t nb = grk_amd_node_encode_image(node, &im, &base, px, GRK_AMD_CS_TLM | GRK_AMD_NODE_GATHER, b, cap);
    /* the same image through one context */
    uint8_t* c = (uint8_t*)malloc(cap);
    const int64_t nc = grk_amd_encode_image(grk_amd_node_ctx(node, 0), &im, &base, px, GRK_AMD_CS_TLM, c, cap);
    printf("devices %u  parallel writers %lld bytes  gather %lld bytes  one context %lld bytes\n", grk_amd_node_size(node),
           (long long)na, (long long)nb, (long long)nc);
    const int ok = na > 0 && na == nb && na == nc && memcmp(a, b, (size_t)na) == 0 && memcmp(a, c, (size_t)na) == 0;
    printf(ok ? "identical\n" : "DIFFERENT\n");
    grk_amd_host_free(grk_amd_node_ctx(node, 0), px);
    grk_amd_node_destroy(node);
    free(a); free(b); free(c);
    return ok ? 0 : 2;
}
