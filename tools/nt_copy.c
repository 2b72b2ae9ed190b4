// tools only: copy with non-temporal stores (the destination lines go to memory, not into this core's cache)
#include <emmintrin.h>
#include <stddef.h>
#include <string.h>
void nt_copy(void* dst, const void* src, size_t n) {
    if (((size_t)dst & 15) || (n & 63)) { memcpy(dst, src, n); return; }
    __m128i* d = (__m128i*)dst;
    const __m128i* s = (const __m128i*)src;
    for (size_t i = 0; i < n / 16; i += 4) {
        __m128i a = _mm_loadu_si128(s + i), b = _mm_loadu_si128(s + i + 1), c = _mm_loadu_si128(s + i + 2), e = _mm_loadu_si128(s + i + 3);
        _mm_stream_si128(d + i, a); _mm_stream_si128(d + i + 1, b); _mm_stream_si128(d + i + 2, c); _mm_stream_si128(d + i + 3, e);
    }
    _mm_sfence();
}
