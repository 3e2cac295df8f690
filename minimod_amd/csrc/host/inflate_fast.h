/* inflate_fast.h -- raw DEFLATE (RFC 1951) decoder for BGZF blocks, whose decoded size is known in advance. */
#ifndef MM_INFLATE_FAST_H
#define MM_INFLATE_FAST_H
#include <stddef.h>
#include <stdint.h>
/* Decodes the raw deflate stream in[0..in_len) into exactly out_len bytes at out.  Returns 0 on success (final block
 * reached, exactly out_len bytes produced), -1 on malformed input or a size mismatch (the caller may then let zlib have
 * a second look, so that error reporting stays zlib's). */
int mm_inflate_raw(const uint8_t *in, size_t in_len, uint8_t *out, size_t out_len);
#endif
