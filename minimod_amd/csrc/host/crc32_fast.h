/* crc32_fast.h -- CRC-32 of a BGZF block payload (see crc32_fast.c) */
#ifndef MM_CRC32_FAST_H
#define MM_CRC32_FAST_H
#include <stddef.h>
#include <stdint.h>
/* the same value as zlib's crc32(crc32(0, NULL, 0), buf, len) */
uint32_t mm_crc32(const uint8_t *buf, size_t len);
#endif
