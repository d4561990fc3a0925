/* Hand-written stand-in for the header the reference's CMake would generate from
 * src/lib/jp2/grk_config.h.cmake.in (values for Grok 8.0.2, CMakeLists.txt:18-22).
 * Test infrastructure only: used by oracle/Makefile to build oracle/_ref. */
#pragma once
#define GRK_VERSION_MAJOR 8
#define GRK_VERSION_MINOR 0
#define GRK_VERSION_BUILD 2
#define GROK_PLUGIN_NAME "grokj2k_plugin"
#define AVX2_FOUND "1"
#define AVX_FOUND "1"
#define SSE4_1_FOUND "1"
#define SSE3_FOUND "1"
