// On-device text bridge: BERT WordPiece ids -> CLIP byte-level BPE ids, no host strings.
//
// Replaces, per step, `tokenizer.batch_decode(topk_inp_batch, skip_special_tokens=True)`
// (gen_utils.py:75) followed by `CLIPTokenizer(text_list, padding=True, max_length=77,
// truncation=True)` (clip/clip.py:71-74), i.e. 51,200 Python/Rust string round trips per step at
// B=256, K=200.  One thread per candidate row does, entirely on integers/bytes:
//   1. WordPiece decode with clean-up: specials dropped, '##' pieces glued, a space before every
//      other piece unless the clean-up rules remove it (tokenizers decoders::wordpiece);
//   2. CLIP pre-split  's|'t|'re|'ve|'m|'ll|'d|\p{L}+|\p{N}|[^\s\p{L}\p{N}]+  over per-byte
//      character classes precomputed on the host (pieces are NFC + lower-cased there);
//   3. byte-level BPE with '</w>' suffix: lowest-rank adjacent pair first, leftmost on ties,
//      merges looked up in an open-addressing hash table (L2-resident, 2 MB);
//   4. <|startoftext|> ... <|endoftext|>, truncated to 77.
// It also produces the two sentiment-path extras that only need the row ids: the repeat count
// (control_gen_utils.py:53) and the per-token lexicon sum (stand-in for sentiments_classifer.py:30).
// All scratch (512 B text + 512 B classes + 64 symbols per thread) lives in LDS.
#include "kernels.h"
#include "bridge_hash.h"

namespace czc {

constexpr int BR_THREADS = 64;
constexpr int BR_MAXB = 512;   // == CZC_BRIDGE_MAX_BYTES
constexpr int BR_MAXSYM = 64;  // longest pre-split chunk in bytes
constexpr int BR_LEN = 77;
constexpr int BR_MAXP = 64;    // pieces per row (== CZC_MAX_BERT_LEN)

__device__ __forceinline__ bool merge_lookup(const BridgeDev& bd, int l, int r, unsigned& rank, int& out) {
  const unsigned long long key = ((unsigned long long)(unsigned)l << 32) | (unsigned)r;
  unsigned h = bridge_hash(key) & bd.hmask;
  for (;;) {
    const unsigned long long k = bd.hkeys[h];
    if (k == key) {
      const unsigned long long v = bd.hvals[h];
      rank = (unsigned)(v >> 32);
      out = (int)(v & 0xFFFFFFFFu);
      return true;
    }
    if (k == ~0ull) return false;
    h = (h + 1) & bd.hmask;
  }
}

__global__ __launch_bounds__(BR_THREADS) void bridge_kernel(BridgeDev bd, const int* inp, int B, int T, int gen_idx,
                                                            const int* cand, int K, const float* lexicon, const float* lex_pos,
                                                            const uint8_t* lex_cls, int negative, PosDev pos, int* clip_ids, int* clip_len, float* senti_raw,
                                                            float* repeats, int* overflow) {
  extern __shared__ __attribute__((aligned(16))) unsigned char br_lds[];
  unsigned char* txt = br_lds + (size_t)threadIdx.x * BR_MAXB;
  unsigned char* cls = br_lds + (size_t)BR_THREADS * BR_MAXB + (size_t)threadIdx.x * BR_MAXB;
  int* sym = (int*)(br_lds + (size_t)2 * BR_THREADS * BR_MAXB) + threadIdx.x * BR_MAXSYM;
  // piece list of the row (first byte, one past the last byte, token id): a pre-split chunk that is exactly one
  // tabulated piece takes its CLIP ids from bd.tok_bpe instead of running the merge loop
  unsigned char* pl_base = br_lds + (size_t)BR_THREADS * (2 * BR_MAXB + BR_MAXSYM * 4);
  unsigned short* pl_beg = (unsigned short*)pl_base + threadIdx.x * BR_MAXP;
  unsigned short* pl_end = (unsigned short*)pl_base + (BR_THREADS + threadIdx.x) * BR_MAXP;
  int* pl_id = (int*)(pl_base + (size_t)BR_THREADS * BR_MAXP * 4) + threadIdx.x * BR_MAXP;
  int np = 0;

  const long row = (long)blockIdx.x * BR_THREADS + threadIdx.x;
  if (row >= (long)B * K) return;
  const int b = (int)(row / K);
  const int cid = cand ? cand[row] : -1;

  // ---- 1. decode to bytes ------------------------------------------------------------------
  int n = 0;
  bool first = true, ovf = false;
  float senti = 0.f;
  int rep = 0;
  int n_words = 0, pos_ok = 0;  // POS control: words = non-special pieces that do not continue a word
  for (int t = 0; t < T; ++t) {
    const int id = (cand && t == gen_idx) ? cid : inp[b * T + t];
    if (cand && id == cid) ++rep;
    const unsigned fl = bd.piece_flags[id];
    if (fl & 1u) continue;  // special token: skipped
    if (lex_pos) {
      // (word, coarse POS) keyed table -- sentiments_classifer.py:14-30 scores each WORD by the mean of its
      // SentiWordNet synsets under the word's coarse POS class ('' n v a r): the word is addressed by its first
      // piece, the class comes from a per-token tag table (context-free stand-in for nltk.pos_tag)
      if (first || !(fl & 2u)) senti += lex_pos[id * 5 + lex_cls[id]];
    } else if (lexicon) {
      senti += lexicon[id];
    }
    if (pos.tag_of_token && (first || !(fl & 2u))) {
      if (n_words < pos.n) {
        const unsigned m = pos.masks[n_words];
        if (m == 0xFFFFu || (m >> pos.tag_of_token[id]) & 1u) ++pos_ok;
      }
      ++n_words;
    }
    const unsigned o0 = bd.piece_off[id], o1 = bd.piece_off[id + 1];
    int need = (int)(o1 - o0) + 2;
    if (n + need > BR_MAXB) { ovf = true; break; }
    if (first) {
      if (fl & 2u) {  // a leading '##' piece keeps its prefix (decode_chain: i == 0 untouched)
        txt[n] = '#'; cls[n] = 2 | 4; ++n;
        txt[n] = '#'; cls[n] = 2 | 4; ++n;
      }
    } else if (!(fl & 6u)) {
      txt[n] = ' '; cls[n] = 3 | 4; ++n;
    }
    if (np < BR_MAXP) { pl_beg[np] = (unsigned short)n; pl_end[np] = (unsigned short)(n + (int)(o1 - o0)); pl_id[np] = id; ++np; }
    for (unsigned o = o0; o < o1; ++o) {
      txt[n] = bd.piece_bytes[o];
      cls[n] = bd.piece_class[o];
      ++n;
    }
    first = false;
  }

  // ---- 2-4. pre-split, BPE, emit -------------------------------------------------------------
  int* outp = clip_ids + row * BR_LEN;
  int nt = 0;  // text tokens emitted (cap 75)
  outp[0] = bd.bos_id;
  int i = 0, pp = 0;
  while (i < n && nt < BR_LEN - 2) {
    const int c = cls[i] & 3;
    if (c == 3) { ++i; continue; }
    int j = 0;
    if (txt[i] == '\'' && i + 1 < n) {
      const unsigned char c1 = txt[i + 1];
      if (c1 == 's' || c1 == 't' || c1 == 'm' || c1 == 'd') j = i + 2;
      else if (i + 2 < n) {
        const unsigned char c2 = txt[i + 2];
        if ((c1 == 'r' && c2 == 'e') || (c1 == 'v' && c2 == 'e') || (c1 == 'l' && c2 == 'l')) j = i + 3;
      }
    }
    if (j == 0) {
      j = i + 1;
      if (c == 0) { while (j < n && (cls[j] & 3) == 0) ++j; }
      else if (c == 1) { while (j < n && !(cls[j] & 4)) ++j; }          // exactly one \p{N} character
      else { while (j < n && (cls[j] & 3) == 2) ++j; }
    }
    if (bd.tok_bpe_len && c == 0) {  // a run of letters that is exactly one tabulated piece: its ids are known
      while (pp < np && pl_beg[pp] < i) ++pp;
      if (pp < np && pl_beg[pp] == i && pl_end[pp] == j) {
        const int id = pl_id[pp];
        const int tl = bd.tok_bpe_len[id];
        if (tl) {
          for (int q = 0; q < tl && nt < BR_LEN - 2; ++q) outp[1 + nt++] = bd.tok_bpe[id * BR_TOKMAX + q];
          i = j;
          continue;
        }
      }
    }
    int m = j - i;
    if (m > BR_MAXSYM) { ovf = true; m = BR_MAXSYM; }
    for (int q = 0; q < m; ++q) sym[q] = bd.byte_sym[txt[i + q]];
    sym[m - 1] = bd.byte_sym_eow[txt[i + m - 1]];
    // BPE: merge the lowest-ranked adjacent pair (leftmost on ties) until none applies
    while (m > 1) {
      unsigned best = 0xFFFFFFFFu;
      int bi = -1, bo = 0;
      for (int q = 0; q + 1 < m; ++q) {
        unsigned rk; int o;
        if (merge_lookup(bd, sym[q], sym[q + 1], rk, o) && rk < best) { best = rk; bi = q; bo = o; }
      }
      if (bi < 0) break;
      sym[bi] = bo;
      for (int q = bi + 1; q + 1 < m; ++q) sym[q] = sym[q + 1];
      --m;
    }
    for (int q = 0; q < m && nt < BR_LEN - 2; ++q) outp[1 + nt++] = sym[q];
    i = j;
  }
  outp[1 + nt] = bd.eos_id;
  for (int q = nt + 2; q < BR_LEN; ++q) outp[q] = bd.eos_id;
  clip_len[row] = nt + 2;
  if (pos.tag_of_token) {
    // template positions past the last word carry the "" padding tag (POS_classifier.py:19-20): it matches the ""
    // wildcard and any entry that accepts it (bit 15: a string entry, `"" in "NOUN"`, or a list that holds "")
    for (int w = n_words; w < pos.n; ++w) pos_ok += (pos.masks[w] & 0x8000u) ? 1 : 0;
    if (senti_raw) senti_raw[row] = (float)pos_ok / (float)pos.n;
  } else if (senti_raw) {
    senti_raw[row] = negative ? -senti : senti;
  }
  if (repeats) repeats[row] = (float)(rep - 1);
  if (ovf) atomicAdd(overflow, 1);
}

// One thread per BERT token: the byte-level BPE of the token standing alone as a word (the same symbol / merge rules as
// above), tabulated when the piece is all letters (so that it is one pre-split chunk whenever spaces surround it) and
// yields at most BR_TOKMAX ids.
__global__ void bridge_precompute_kernel(BridgeDev bd, int* tok_ids, uint8_t* tok_len) {
  const int id = blockIdx.x * blockDim.x + threadIdx.x;
  if (id >= bd.bert_vocab) return;
  tok_len[id] = 0;
  const unsigned o0 = bd.piece_off[id], o1 = bd.piece_off[id + 1];
  int m = (int)(o1 - o0);
  if (m <= 0 || m > BR_MAXSYM || (bd.piece_flags[id] & 1u)) return;
  for (unsigned o = o0; o < o1; ++o)
    if ((bd.piece_class[o] & 3) != 0) return;
  int sym[BR_MAXSYM];
  for (int q = 0; q < m; ++q) sym[q] = bd.byte_sym[bd.piece_bytes[o0 + q]];
  sym[m - 1] = bd.byte_sym_eow[bd.piece_bytes[o0 + m - 1]];
  while (m > 1) {
    unsigned best = 0xFFFFFFFFu;
    int bi = -1, bo = 0;
    for (int q = 0; q + 1 < m; ++q) {
      unsigned rk; int o;
      if (merge_lookup(bd, sym[q], sym[q + 1], rk, o) && rk < best) { best = rk; bi = q; bo = o; }
    }
    if (bi < 0) break;
    sym[bi] = bo;
    for (int q = bi + 1; q + 1 < m; ++q) sym[q] = sym[q + 1];
    --m;
  }
  if (m > BR_TOKMAX) return;
  for (int q = 0; q < m; ++q) tok_ids[id * BR_TOKMAX + q] = sym[q];
  tok_len[id] = (uint8_t)m;
}

int launch_bridge_precompute(const BridgeDev& bd, int* tok_ids, uint8_t* tok_len, hipStream_t st) {
  hipLaunchKernelGGL(bridge_precompute_kernel, dim3(cdiv(bd.bert_vocab, 64)), dim3(64), 0, st, bd, tok_ids, tok_len);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

int launch_bridge(const BridgeDev& bd, const int* inp, int B, int T, int gen_idx, const int* cand, int K,
                  const float* lexicon, const float* lex_pos, const uint8_t* lex_cls, int negative, const PosDev& pos, int* clip_ids, int* clip_len, float* senti_raw,
                  float* repeats, int* overflow_flag, hipStream_t st) {
  const long rows = (long)B * K;
  if (rows <= 0) return 0;
  const size_t shmem = (size_t)BR_THREADS * (2 * BR_MAXB + BR_MAXSYM * 4 + BR_MAXP * 8);
  CZC_HIP_CHECK(hipFuncSetAttribute((const void*)bridge_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)shmem));
  hipLaunchKernelGGL(bridge_kernel, dim3(cdiv(rows, BR_THREADS)), dim3(BR_THREADS), shmem, st, bd, inp, B, T, gen_idx,
                     cand, K, lexicon, lex_pos, lex_cls, negative, pos, clip_ids, clip_len, senti_raw, repeats, overflow_flag);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- exclusive scan of sequence lengths --------------------------------------------------------
__global__ __launch_bounds__(1024) void scan_kernel(const int* len, int n, int* off, int* totals) {
  __shared__ int part[1024];
  __shared__ int pmax[1024];
  const int tid = threadIdx.x;
  const int per = (n + 1023) / 1024;
  const int lo = tid * per, hi = min(n, lo + per);
  int s = 0, mx = 0;
  for (int i = lo; i < hi; ++i) { s += len[i]; mx = max(mx, len[i]); }
  part[tid] = s;
  pmax[tid] = mx;
  __syncthreads();
  for (int o = 1; o < 1024; o <<= 1) {
    const int v = tid >= o ? part[tid - o] : 0;
    const int w = tid >= o ? pmax[tid - o] : 0;
    __syncthreads();
    part[tid] += v;
    pmax[tid] = max(pmax[tid], w);
    __syncthreads();
  }
  int run = part[tid] - s;
  for (int i = lo; i < hi; ++i) { off[i] = run; run += len[i]; }
  if (tid == 1023) { off[n] = part[1023]; totals[0] = part[1023]; totals[1] = pmax[1023]; }
}

int launch_scan(const int* len, int n, int* off, int* totals, hipStream_t st) {
  hipLaunchKernelGGL(scan_kernel, dim3(1), dim3(1024), 0, st, len, n, off, totals);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- shared-prefix plan ---------------------------------------------------------------------------
// Causal attention + EOS pooling make every hidden state in front of the first differing CLIP token
// identical across the K candidates of an image, so that prefix is encoded once.
//
// Exact de-duplication (rep != nullptr): candidates of one image whose CLIP id rows are identical -- every candidate the
// token mask turned into [PAD] decodes to the SAME caption without the word (gen_utils.py:72-75; with a real stop-word
// mask at tau = 0.1 that is most of the tail of the K candidates), and different word pieces can decode to the same string --
// have identical hidden states in every row.  The first of them (lowest k) keeps its rows; the others get NO rows
// (own_len = 0) and point their EOS index at the representative's EOS row (prefix_finish_kernel), so the pooled feature,
// the cosine and everything behind it are the representative's, bit for bit.  In the f32 engine that is exactly what their
// own rows would have produced; in the MFMA engines the copies of one sentence never agreed among themselves to the last bit
// (a candidate's softmax sum is associated by its slot inside the 32-query attention tile), they now share one of those
// values (tests/test_step_gpu.py::test_dedup_is_exact spells out what is asserted per precision).
// Detection: 64-bit FNV-1a of (length, ids) per candidate in LDS, K^2 / 256 hash compares per thread, every hash match
// confirmed id by id (no false merges).
__global__ __launch_bounds__(256) void prefix_plan_kernel(const int* cids, const int* clen, int B, int K, int share,
                                                          int* own_len, int* pre_len, int* seg_src, int* seg_pos0,
                                                          int* max_len_out, int* img_max, int* rep, int* n_dup_out) {
  __shared__ int s_p, s_max, s_keys, s_dup;
  __shared__ unsigned long long s_hash[1024];  // K <= CZC_MAX_TOPK
  __shared__ short s_len[1024];                 // sequence lengths (<= 77)
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) { s_p = 1 << 30; s_max = 0; s_keys = 0; s_dup = 0; }
  __syncthreads();
  const int* r0 = cids + (long)b * K * BR_LEN;
  const int l0 = clen[b * K];
  int p = 1 << 30, mx = 0;
  for (int k = tid; k < K; k += blockDim.x) {
    const int* rk = r0 + (long)k * BR_LEN;
    const int lk = clen[b * K + k];
    const int lim = min(lk, l0);
    int c = 0;
    while (c < lim && rk[c] == r0[c]) ++c;
    p = min(p, min(c, lk - 1));
    mx = max(mx, lk);
    if (rep) {
      unsigned long long h = 1469598103934665603ull ^ (unsigned)lk;
      for (int c2 = 0; c2 < lk; ++c2) { h ^= (unsigned)rk[c2]; h *= 1099511628211ull; }
      s_hash[k] = h;   // (the length is part of the hash: equal hashes of unequal lengths are caught by the id compare below)
      s_len[k] = (short)lk;
    }
  }
  atomicMin(&s_p, p);
  atomicMax(&s_max, mx);
  __syncthreads();
  const int pb = share ? s_p : 0;
  if (tid == 0) {
    own_len[b] = pb; pre_len[b] = 0; seg_src[b] = b * K; seg_pos0[b] = 0;
    atomicMax(max_len_out, s_max);
    atomicMax(max_len_out + 1, s_max - pb);  // longest branch (own rows of one candidate)
    if (img_max) img_max[b] = s_max - pb;
  }
  int keys = 0;  // causal (query, key) pairs of this image's segments: what the attention of one layer and head multiplies
  int dups = 0;
  for (int k = tid; k < K; k += blockDim.x) {
    const int s = B + b * K + k;
    const int lk = clen[b * K + k];
    int own = lk - pb;
    if (rep) {
      int r = k;
      const unsigned long long hk = s_hash[k];
      const int* rk = r0 + (long)k * BR_LEN;
      for (int j = 0; j < k; ++j) {          // LDS only until a hash matches (K^2 / 512 compares per thread)
        if (s_hash[j] != hk || s_len[j] != lk) continue;
        const int* rj = r0 + (long)j * BR_LEN;
        int c = pb;                            // the first pb ids are the image's common prefix: equal by construction
        while (c < lk && rj[c] == rk[c]) ++c;
        if (c == lk) { r = j; break; }  // the lowest identical candidate is the representative (it is its own: rep[j] == j)
      }
      rep[b * K + k] = r;
      if (r != k) { own = 0; ++dups; }
    }
    own_len[s] = own;
    pre_len[s] = pb;
    seg_src[s] = b * K + k;
    seg_pos0[s] = pb;
    keys += own * pb + own * (own + 1) / 2;
  }
  atomicAdd(&s_keys, keys);
  if (dups) atomicAdd(&s_dup, dups);
  __syncthreads();
  if (tid == 0) {
    atomicAdd(max_len_out + 4, s_keys + pb * (pb + 1) / 2);  // totals[7] (profiling: attention FLOPs)
    if (n_dup_out && s_dup) atomicAdd(n_dup_out, s_dup);     // totals[5]: candidates that ride on another one's rows
  }
}

int launch_prefix_plan(const int* clip_ids, const int* clip_len, int B, int K, int share, int* own_len, int* pre_len,
                       int* seg_src, int* seg_pos0, int* max_len_out, int* img_max, hipStream_t st, int* rep, int* n_dup_out) {
  if (rep && K > 1024) { snprintf(g_err, sizeof(g_err), "prefix plan: de-duplication needs K <= 1024"); return 1; }
  hipLaunchKernelGGL(prefix_plan_kernel, dim3(B), dim3(256), 0, st, clip_ids, clip_len, B, K, share, own_len, pre_len,
                     seg_src, seg_pos0, max_len_out, img_max, rep, n_dup_out);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// rep (optional, [B*K]): candidate i pools the EOS row of candidate rep[i] of the same image (its own when rep[i] == i % K)
__global__ void prefix_finish_kernel(const int* own_off, const int* own_len, int B, int K, int* pre_off, int* eos_idx,
                                     int* n_trunk_rows, const int* rep) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i == 0 && n_trunk_rows) *n_trunk_rows = own_off[B];  // the B trunk segments come first
  if (i < B) pre_off[i] = 0;
  if (i < B * K) {
    const int s = B + i;
    pre_off[s] = own_off[i / K];
    const int sr = rep ? B + (i / K) * K + rep[i] : s;
    eos_idx[i] = own_off[sr] + own_len[sr] - 1;
  }
}

int launch_prefix_finish(const int* own_off, const int* own_len, int B, int K, int* pre_off, int* eos_idx,
                         int* n_trunk_rows, hipStream_t st, const int* rep) {
  hipLaunchKernelGGL(prefix_finish_kernel, dim3(cdiv((long)B * K, 256)), dim3(256), 0, st, own_off, own_len, B, K,
                     pre_off, eos_idx, n_trunk_rows, rep);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// ---- screen-then-refine: segment plan of the second (split-fp16) pass ---------------------------------
// One work-group per image.  count_off = exclusive scan of count (count_off[B] = R), *kr = max_b count[b].  The plan
// keeps the REGULAR shape of the screening plan -- B trunks, then B x Kr branch slots, image b's chosen candidates in
// slots b*Kr .. b*Kr + count[b] - 1 and the rest of its slots empty (own_len 0, pre-zeroed by the caller) -- so that the
// packed-branch attention kernels serve it; pooled rows / rlist stay compact (row r = count_off[b] + i).  Trunk b keeps
// the prefix length of the screening plan.
__global__ __launch_bounds__(256) void refine_plan_kernel(const int* clen, const int* trunk_len, const int* list, const int* count,
                                                          const int* count_off, const int* kr, int B, int K, int* own_len, int* pre_len,
                                                          int* seg_src, int* seg_pos0, int* rlist, int* max_len_out, int* img_max) {
  __shared__ int s_mxb;
  const int b = blockIdx.x, tid = threadIdx.x;
  if (tid == 0) s_mxb = 0;
  __syncthreads();
  const int n = count[b], o = count_off[b], Kr = *kr;
  const int pb = n > 0 ? trunk_len[b] : 0;
  if (tid == 0) { own_len[b] = pb; pre_len[b] = 0; seg_src[b] = b * K; seg_pos0[b] = 0; }
  int mx = 0, mxb = 0;
  for (int i = tid; i < Kr; i += blockDim.x) {
    const int s = B + b * Kr + i;
    if (i >= n) {  // empty slot: a zero-length segment of this image's trunk
      pre_len[s] = pb; seg_src[s] = b * K; seg_pos0[s] = pb;
      continue;
    }
    const int flat = b * K + list[(long)b * K + i];
    const int len = clen[flat];
    own_len[s] = len - pb;
    pre_len[s] = pb;
    seg_src[s] = flat;
    seg_pos0[s] = pb;
    rlist[o + i] = flat;
    mx = max(mx, len);
    mxb = max(mxb, len - pb);
    atomicAdd(max_len_out + 4, (len - pb) * pb + (len - pb) * (len - pb + 1) / 2);  // rtot[7]: (query, key) pairs
  }
  if (tid == 0 && n > 0) atomicAdd(max_len_out + 4, pb * (pb + 1) / 2);
  if (mx) { atomicMax(max_len_out, mx); atomicMax(max_len_out + 1, mxb); atomicMax(&s_mxb, mxb); }
  __syncthreads();
  if (tid == 0 && img_max) img_max[b] = s_mxb;
}

int launch_refine_plan(const int* clip_len, const int* trunk_len, const int* list, const int* count, const int* count_off,
                       const int* kr_dev, int B, int K, int* own_len, int* pre_len, int* seg_src, int* seg_pos0, int* rlist,
                       int* max_len_out, int* img_max, hipStream_t st) {
  hipLaunchKernelGGL(refine_plan_kernel, dim3(B), dim3(256), 0, st, clip_len, trunk_len, list, count, count_off, kr_dev, B, K, own_len,
                     pre_len, seg_src, seg_pos0, rlist, max_len_out, img_max);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

// after the scan of own_len over B + B*K segment slots: pre_off of every branch slot = own_off[its image's trunk];
// eos_idx[r] of the compact row r = count_off[b] + i
__global__ void refine_finish_kernel(const int* own_off, const int* own_len, const int* count, const int* count_off, const int* kr,
                                     int B, int* pre_off, int* eos_idx) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int Kr = *kr;
  if (i < B) pre_off[i] = 0;
  if (i < B * Kr) {
    const int b = i / Kr, j = i - b * Kr;
    const int s = B + i;
    pre_off[s] = own_off[b];
    if (j < count[b]) eos_idx[count_off[b] + j] = own_off[s] + own_len[s] - 1;
  }
}

int launch_refine_finish(const int* own_off, const int* own_len, const int* count, const int* count_off, const int* kr_dev, int B, int K,
                         int* pre_off, int* eos_idx, hipStream_t st) {
  hipLaunchKernelGGL(refine_finish_kernel, dim3(cdiv((long)B * K, 256)), dim3(256), 0, st, own_off, own_len, count, count_off, kr_dev, B,
                     pre_off, eos_idx);
  CZC_HIP_CHECK(hipGetLastError());
  return 0;
}

}  // namespace czc
