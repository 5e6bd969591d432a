/* TEST INFRASTRUCTURE ONLY -- CPU model of the device-side LZ4 block compressor (cloudini_amd/csrc/lz4_kernels.hip).
 *
 * stage 2 on the device need not reproduce LZ4_compress_default's bytes (lz4 v1.10.0,
 * the reference's call site is src/codec_common.cpp:232-234); what it must produce is a valid LZ4 *block* that
 * LZ4_decompress_safe (src/codec_common.cpp:275) turns back into the exact stage-1 payload. The block format is
 * the published one (lz4_Block_format.md): sequences of [token][literal length bytes][literals][offset u16 LE]
 * [match length bytes], a last sequence of literals only, the last 5 bytes literals, no match starting within the last
 * 12 bytes.
 *
 * The device algorithm is deterministic, and this file restates it serially so that the GPU tests can ask for
 * byte equality (and the CPU tests for validity against the system's liblz4):
 *   - the payload is cut into sub-ranges of `sub_bytes` (8192 on the device, 2048-entry table, 1024 matches); a sub-range is parsed by one wave with its own hash table
 *     (2^hash_bits entries, position inside the sub-range + 1), so matches never leave their sub-range;
 *   - the wave looks at 64 consecutive positions per step: every lane hashes the 4 bytes at its position, checks the
 *     table's candidate (the table as it was before the step) and, on a hit, extends its match as far as it goes (up to
 *     the sub-range end and the block's end rules). All 64 positions are then entered into the table (the highest
 *     position wins a slot). The step's matches are taken greedily in position order: the first hit, then the first hit
 *     at or behind its end, and so on. The cursor moves by 64, or behind the last match taken if that reaches further;
 *   - a step yields 16 matches at most (they do not overlap, 4 bytes or more each), and a step is only parsed while the
 *     list of `max_matches` has room for 16 (the rest of the sub-range is literals);
 *   - sequences are then emitted over the whole block from the ordered list of matches.
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef struct {
  uint32_t pos, len, off;
} lz4m_match;

static uint32_t rd32(const uint8_t* p) {
  uint32_t v;
  memcpy(&v, p, 4);
  return v;
}

static uint32_t ext_bytes(uint32_t x) { return x >= 15u ? (x - 15u) / 255u + 1u : 0u; }

static uint8_t* put_ext(uint8_t* o, uint32_t x) { /* x >= 15: the part above 15 as 255, 255, ..., rest */
  x -= 15u;
  while (x >= 255u) {
    *o++ = 255u;
    x -= 255u;
  }
  *o++ = (uint8_t)x;
  return o;
}

/* matches of sub-range [s, e) of the n-byte block, appended to m[] (capacity max_matches); returns their number */
static uint32_t lz4m_parse(const uint8_t* in, uint32_t n, uint32_t s, uint32_t e, uint32_t hash_bits, uint32_t max_matches,
                           uint16_t* table, lz4m_match* m) {
  memset(table, 0, sizeof(uint16_t) << hash_bits);
  if (n < 13u || e - s < 4u) return 0u;
  const int64_t last_start = (int64_t)(e - 4u) < (int64_t)n - 12 ? (int64_t)(e - 4u) : (int64_t)n - 12;
  const uint32_t end_limit = e < n - 5u ? e : n - 5u; /* a match ends here at the latest */
  uint32_t count = 0u;
  int64_t i = s;
  while (i <= last_start && count + 16u <= max_matches) {
    uint32_t len[64], cf[64];
    for (int l = 0; l < 64; ++l) { /* every lane looks at the table as it was BEFORE the step */
      const int64_t p = i + l;
      len[l] = 0u;
      if (p > last_start) continue;
      const uint32_t seq = rd32(in + p);
      const uint32_t h = (seq * 2654435761u) >> (32u - hash_bits);
      const uint32_t cand = table[h];
      if (cand != 0u && rd32(in + s + cand - 1u) == seq) {
        const uint32_t c = s + cand - 1u, maxlen = end_limit - (uint32_t)p;
        uint32_t k = 4u;
        while (k < maxlen && in[p + k] == in[c + k]) ++k;
        len[l] = k;
        cf[l] = c;
      }
    }
    for (int l = 0; l < 64; ++l) { /* ascending positions: within a step the highest position wins a slot */
      const int64_t p = i + l;
      if (p > last_start) break;
      const uint32_t h = (rd32(in + p) * 2654435761u) >> (32u - hash_bits);
      table[h] = (uint16_t)(p - s + 1);
    }
    uint32_t cur = 0u;
    for (uint32_t l = 0; l < 64u; ++l) {
      if (l < cur || len[l] == 0u) continue;
      m[count].pos = (uint32_t)(i + l);
      m[count].len = len[l];
      m[count].off = (uint32_t)(i + l) - cf[l];
      ++count;
      cur = l + len[l];
    }
    i += cur > 64u ? cur : 64u;
  }
  return count;
}

/* worst-case size of the block this compressor writes for n input bytes (no match at all) */
uint32_t orc_lz4_bound(uint32_t n) { return n + n / 255u + 16u; }

/* returns the block size, or -1 when `cap` is too small / the parameters are out of range */
int64_t orc_lz4_compress(const uint8_t* in, uint32_t n, uint8_t* out, uint32_t cap, uint32_t sub_bytes, uint32_t hash_bits,
                         uint32_t max_matches) {
  if (sub_bytes < 64u || sub_bytes > 65535u || hash_bits < 4u || hash_bits > 16u || max_matches == 0u) return -1;
  if (cap < orc_lz4_bound(n)) return -1;
  uint16_t* table = (uint16_t*)malloc(sizeof(uint16_t) << hash_bits);
  lz4m_match* m = (lz4m_match*)malloc(sizeof(lz4m_match) * (size_t)max_matches);
  if (!table || !m) {
    free(table);
    free(m);
    return -1;
  }
  uint8_t* o = out;
  uint32_t anchor = 0u;
  for (uint32_t s = 0u; s < n; s += sub_bytes) {
    const uint32_t e = n - s < sub_bytes ? n : s + sub_bytes;
    const uint32_t cnt = lz4m_parse(in, n, s, e, hash_bits, max_matches, table, m);
    for (uint32_t k = 0; k < cnt; ++k) {
      const uint32_t lit = m[k].pos - anchor, ml = m[k].len - 4u;
      *o++ = (uint8_t)(((lit < 15u ? lit : 15u) << 4) | (ml < 15u ? ml : 15u));
      if (lit >= 15u) o = put_ext(o, lit);
      memcpy(o, in + anchor, lit);
      o += lit;
      *o++ = (uint8_t)(m[k].off & 0xffu);
      *o++ = (uint8_t)(m[k].off >> 8);
      if (ml >= 15u) o = put_ext(o, ml);
      anchor = m[k].pos + m[k].len;
    }
  }
  { /* last sequence: literals only */
    const uint32_t lit = n - anchor;
    *o++ = (uint8_t)((lit < 15u ? lit : 15u) << 4);
    if (lit >= 15u) o = put_ext(o, lit);
    memcpy(o, in + anchor, lit);
    o += lit;
  }
  free(table);
  free(m);
  (void)ext_bytes;
  return (int64_t)(o - out);
}
