// stage1_decode_automaton.h -- k_mark_ends_automaton (round 5): the token ends of regular streams with RAW fields between
// the varints (FieldDecoderCopy / FieldDecoderFloat_XOR next to varint-coded fields, include/cloudini_lib/field_decoder.hpp:
// 56-75, 108-156; decodeVarint, encoding_utils.hpp:98-148), found as the run of a finite automaton over the bytes.
//
// A raw byte may look like anything, so the ends of the varints cannot be read off the MSBs; what is known is the FORM of a
// point: op after op, a varint or `size` raw bytes. That form is an automaton whose states are the places inside a point --
// one state per varint op ("inside varint j"), one per byte of a raw field -- numbered in form order:
//     varint state:  a byte with its MSB set keeps the state, any other byte moves on and ENDS a token;
//     raw state:     every byte moves on; the field's last byte ends a token;        the last state moves on to state 0.
// There are only two transition maps (MSB set / clear). A map of S <= 16 states is 16 four-bit fields of one 64-bit word,
// and maps compose, so the state in front of every byte follows from a prefix "sum" under composition:
//   * a lane takes 32 consecutive payload bytes: their 32 MSBs, eight look-ups per start state in a 256-byte LDS table
//     (T4[MSB nibble][state] = state behind four bytes | their end bits << 4) give the lane's own map;
//   * one DPP scan composes the maps of the 64 lanes of a piece (2 KiB); the piece's map is lane 63's;
//   * the state behind piece p travels from wave to wave through tagged LDS records (one look-up per hop), like the token
//     counts of k_decode_stream_w;
//   * knowing its incoming state a lane walks its eight groups once more and has its 32 end bits: ONE word of the chunk's
//     end bitmap (one bit per payload byte, token_ends_word), stored coalesced.
// k_decode_stream_w then runs in bitmap mode as behind k_mark_token_ends -- which this kernel replaces wherever the form has
// at most 16 states (rounds 3-4: pointer doubling over 1 KiB tiles, 3.5 ms per 16 M XYZ + packed rgb points; the FORM mode
// of the stream kernel, 1.3 ms). Varints of more than 10 bytes and streams that end inside a point are not judged here: the
// stream kernel hands such chunks to the serial decoder, as in MSB mode.
// grid = n_chunks, kMaWaves * 64 threads.
#pragma once

namespace cldn {

constexpr uint32_t kMaWaves = 8u;
constexpr uint32_t kMaPiece = 2048u;  // payload bytes per piece: 32 per lane
constexpr uint32_t kMaRing = 64u;
constexpr uint32_t kMaMaxStates = 16u;
constexpr uint32_t kMaSpinLimit = 1u << 20;

// states of the form (0 when it has more than kMaMaxStates)
inline uint32_t automaton_states(const DevPlan& P) {
  uint32_t s = 0u;
  for (uint32_t o = 0; o < P.n_ops; ++o) {
    const uint32_t kd = P.ops[o].kind;
    s += (kd == OP_COPY || kd == OP_XOR32 || kd == OP_XOR64) ? P.ops[o].size : 1u;
  }
  return s <= kMaMaxStates ? s : 0u;
}

// (a o b)(s) = a(b(s)): b first. Fields beyond S are never looked at.
__device__ __forceinline__ uint64_t ma_compose(uint64_t a, uint64_t b, uint32_t S) {
  uint64_t r = 0ull;
  for (uint32_t s = 0; s < S; ++s) {  // uniform
    const uint32_t t = (uint32_t)(b >> (4u * s)) & 15u;
    r |= ((a >> (4u * t)) & 15ull) << (4u * s);
  }
  return r;
}

// ... of at most 8 states: eight 4-bit fields of one dword (v_bfe_u32 with a register offset does the look-up)
__device__ __forceinline__ uint32_t ma_compose(uint32_t a, uint32_t b, uint32_t S) {
  uint32_t r = 0u;
  for (uint32_t s = 0; s < S; ++s) {  // uniform
    const uint32_t t = (b >> (4u * s)) & 15u;
    r |= ((a >> (4u * t)) & 15u) << (4u * s);
  }
  return r;
}
#define MA_DPP32(X, CTRL, RMASK, BC, OLD) ((uint32_t)__builtin_amdgcn_update_dpp((int)(OLD), (int)(X), CTRL, RMASK, 0xf, BC))
#define MA_DPP64(X, CTRL, RMASK, BC, OLD)                                                                                  \
  ((((uint64_t)(uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)((OLD) >> 32), (int)(uint32_t)((X) >> 32), CTRL, RMASK, 0xf, BC)) << 32) | \
   (uint64_t)(uint32_t)__builtin_amdgcn_update_dpp((int)(uint32_t)(OLD), (int)(uint32_t)(X), CTRL, RMASK, 0xf, BC))

// MAP = uint32_t: forms of at most 8 states; uint64_t: of at most 16
template <typename MAP>
__global__ __launch_bounds__(kMaWaves * 64) void k_mark_ends_automaton(const DevPlan plan, const uint8_t* __restrict__ streams,
                                                                       const DecChunk* __restrict__ chunks,
                                                                       uint32_t* __restrict__ token_ends, uint32_t* __restrict__ reg_end) {
  __shared__ uint8_t T1[2][16];     // one byte: [MSB][state] -> next state | end << 4
  __shared__ uint8_t T4[16][16];    // four bytes: [MSB nibble, byte 0 in bit 0][state] -> state behind them | end bits << 4
  __shared__ unsigned long long rec[kMaRing];  // {tag = piece + 1, state behind the piece}
  __shared__ uint32_t gave_up;
  const uint32_t c = blockIdx.x;
  const uint32_t tid = threadIdx.x, lane = tid & 63u;
  const uint32_t wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const DecChunk dc = chunks[c];
  if (!dc.valid) return;
  const uint8_t* src = streams + dc.src_off;
  const uint32_t src_size = dc.src_size;
  uint32_t* out_words = token_ends + token_ends_word(dc.src_off, c);

  // ---- the automaton of the plan's point form
  uint32_t S = 0u;  // number of states (uniform)
  {
    // thread t < 32: state t & 15, MSB t >> 4: walk the ops to find which op the state belongs to (every thread walks: S)
    const uint32_t st = tid & 15u, msb = (tid >> 4) & 1u;
    uint32_t first = 0u, nxt = 0u, end = 0u;
    bool found = false;
    for (uint32_t o = 0; o < plan.n_ops; ++o) {
      const uint32_t kd = plan.ops[o].kind;
      const bool raw = kd == OP_COPY || kd == OP_XOR32 || kd == OP_XOR64;
      const uint32_t ns = raw ? plan.ops[o].size : 1u;
      if (!found && st >= first && st < first + ns) {
        found = true;
        if (raw) {
          nxt = st + 1u;
          end = st + 1u == first + ns ? 1u : 0u;
        } else {
          nxt = msb ? st : st + 1u;
          end = msb ? 0u : 1u;
        }
      }
      first += ns;
    }
    S = first;
    if (nxt == S) nxt = 0u;        // behind the last op: the next point
    if (!found) {                  // (states beyond S are never entered)
      nxt = 0u;
      end = 0u;
    }
    if (tid < 32u) T1[msb][st] = (uint8_t)(nxt | (end << 4));
  }
  S = (uint32_t)__builtin_amdgcn_readfirstlane((int)S);
  for (uint32_t i = tid; i < kMaRing; i += kMaWaves * 64u) rec[i] = 0ull;
  if (tid == 0) {
    gave_up = 0u;
    reg_end[c] = 0u;
  }
  __syncthreads();
  if (tid < 256u) {
    const uint32_t nib = tid >> 4;
    uint32_t st = tid & 15u, ends = 0u;
    for (uint32_t b = 0; b < 4u; ++b) {
      const uint32_t e = T1[(nib >> b) & 1u][st];
      st = e & 15u;
      ends |= (e >> 4) << b;
    }
    T4[nib][tid & 15u] = (uint8_t)(st | (ends << 4));
  }
  __syncthreads();
  const uint8_t* t4 = &T4[0][0];

  const uint32_t n_pieces = (src_size + kMaPiece - 1u) / kMaPiece;
  constexpr bool W64 = sizeof(MAP) == 8;
  MAP ident = 0;
  for (uint32_t s = 0; s < (W64 ? 16u : 8u); ++s) ident |= (MAP)s << (4u * s);

  for (uint32_t p = wave; p < n_pieces; p += kMaWaves) {
    // ---- my 32 bytes (the lane that holds the payload's end reads them one by one; lanes behind it read nothing)
    const uint32_t b0 = p * kMaPiece + lane * 32u;
    uint32_t w[8];
    if (b0 + 32u <= src_size) {
      uint4 u0, u1;
      __builtin_memcpy(&u0, src + b0, 16);
      __builtin_memcpy(&u1, src + b0 + 16u, 16);
      w[0] = u0.x; w[1] = u0.y; w[2] = u0.z; w[3] = u0.w;
      w[4] = u1.x; w[5] = u1.y; w[6] = u1.z; w[7] = u1.w;
    } else {
#pragma unroll
      for (int k = 0; k < 8; ++k) {
        w[k] = 0u;
        for (uint32_t bb = 0; bb < 4u; ++bb) {
          const uint32_t at = b0 + 4u * (uint32_t)k + bb;
          if (at < src_size) w[k] |= (uint32_t)src[at] << (8u * bb);
        }
      }
    }
    // bit i = MSB of byte i: four MSBs per dword by one v_dot4 (weights 1, 2, 4, 8 of 0x80 = the nibble << 7)
    uint32_t msb = 0u;
#pragma unroll
    for (int k = 0; k < 8; ++k) msb |= (__builtin_amdgcn_udot4(w[k] & 0x80808080u, 0x08040201u, 0u, false) >> 7) << (4 * k);

    // ---- my map: where each start state is behind my 32 bytes
    MAP F = 0;
    for (uint32_t s = 0; s < S; ++s) {  // uniform
      uint32_t st = s;
#pragma unroll
      for (int g = 0; g < 8; ++g) st = t4[(((msb >> (4 * g)) & 15u) << 4) | st] & 15u;
      F |= (MAP)st << (4u * s);
    }
    // ---- inclusive scan under composition: X_l = F_l o ... o F_0
    MAP X = F;
#define MA_STEP(CTRL, RMASK, BC)                                    \
  {                                                                 \
    MAP o;                                                          \
    if constexpr (W64) o = MA_DPP64(X, CTRL, RMASK, BC, ident);     \
    else o = MA_DPP32(X, CTRL, RMASK, BC, ident);                   \
    X = ma_compose(X, o, S);                                        \
  }
    MA_STEP(0x111, 0xf, false)   // row_shr:1 (lanes without a source compose with the identity: `old` operand)
    MA_STEP(0x112, 0xf, false)   // row_shr:2
    MA_STEP(0x114, 0xf, false)   // row_shr:4
    MA_STEP(0x118, 0xf, false)   // row_shr:8
    MA_STEP(0x142, 0xa, false)   // row_bcast:15 -> rows 1, 3
    MA_STEP(0x143, 0xc, false)   // row_bcast:31 -> rows 2, 3
#undef MA_STEP
    MAP total, G;  // the piece's map; the map in front of my bytes: lane l - 1's inclusive map, the identity for lane 0
    if constexpr (W64) {
      total = (((uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)(X >> 32), 63)) << 32) |
              (uint64_t)(uint32_t)__builtin_amdgcn_readlane((int)(uint32_t)X, 63);
      G = MA_DPP64(X, 0x138, 0xf, false, ident);  // wave_shr:1
    } else {
      total = (uint32_t)__builtin_amdgcn_readlane((int)X, 63);
      G = MA_DPP32(X, 0x138, 0xf, false, ident);
    }

    // ---- the state in front of the piece: one hop of the chain
    uint32_t s_in = 0u;
    if (p != 0u) {
      const unsigned long long* r = rec + ((p - 1u) & (kMaRing - 1u));
      unsigned long long x = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      for (uint32_t spins = 0; (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)(x >> 32)) != p; ++spins) {
        if (spins >= kMaSpinLimit || *(volatile uint32_t*)&gave_up != 0u) {
          if (lane == 0u) gave_up = 1u;
          break;
        }
        __builtin_amdgcn_s_sleep(2);
        x = __hip_atomic_load(r, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
      }
      s_in = (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)x) & 15u;
    }
    if (*(volatile uint32_t*)&gave_up != 0u) break;  // (uniform per wave: every wave sees it at its next piece at the latest)
    const uint32_t s_out = (uint32_t)(total >> (4u * s_in)) & 15u;
    if (lane == 0u)
      __hip_atomic_store(rec + (p & (kMaRing - 1u)), ((unsigned long long)(p + 1u) << 32) | s_out, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);

    // ---- my end bits
    uint32_t st = (uint32_t)(G >> (4u * s_in)) & 15u, ends = 0u;
#pragma unroll
    for (int g = 0; g < 8; ++g) {
      const uint32_t e = t4[(((msb >> (4 * g)) & 15u) << 4) | st];
      st = e & 15u;
      ends |= (e >> 4) << (4 * g);
    }
    if (b0 < src_size) {
      const uint32_t valid = src_size - b0;
      if (valid < 32u) ends &= (1u << valid) - 1u;
      out_words[p * (kMaPiece / 32u) + lane] = ends;
    }
  }
  __syncthreads();
  if (tid == 0 && gave_up) reg_end[c] = kDecRedo;  // (a wave waited in vain: the serial decoder takes the chunk)
}

}  // namespace cldn
