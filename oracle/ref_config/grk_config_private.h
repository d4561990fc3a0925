/* Hand-written stand-in for grk_config_private.h (see grk_config.h beside it). */
#pragma once
#define GROK_HAVE_INTTYPES_H 1
#define GRK_PACKAGE_VERSION "8.0.2"
#define _LARGEFILE_SOURCE
#define _LARGE_FILES
#define _FILE_OFFSET_BITS 64
#define GROK_HAVE_FSEEKO 1
#define GROK_HAVE_MALLOC_H
#define GROK_HAVE_ALIGNED_ALLOC
#define GROK_HAVE_MEMALIGN
#define GROK_HAVE_POSIX_MEMALIGN
#if !defined(_POSIX_C_SOURCE)
#define _POSIX_C_SOURCE 200112L
#endif
