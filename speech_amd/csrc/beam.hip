// CTC prefix beam search with an n-gram language model.
// Replaces the reference's pyctcdecode path (scripts/eval_dataset.py:20-27 build_ctcdecoder(CHARS,
// kenlm scorer, alpha 0.3, beta 1.0); :43-51 decode_batch over a 16-process pool on CPU logits).
//
// Split the way the work splits on this machine:
//   sp_beam_candidates   GPU.  One wave per frame: log-softmax of the row (max, log-sum-exp), clip to
//                        [log 1e-15, 0], then the frame's candidate tokens -- every token with
//                        log p >= token_min_logp plus the argmax -- compacted in ascending id order by
//                        ballot/popcount.  The scores are read once (HBM-bound, like the greedy argmax)
//                        and only a few (id, log p) pairs per frame cross PCIe instead of T x V floats.
//   sp_ctc_beam_decode   host, native, one thread per utterance (the reference's process pool).  Beams are
//                        (prefix, last token) pairs over a trie of normalised prefixes; a finished word
//                        is scored once, when its boundary node is created.
//   sp_ngram_*           ARPA n-gram model with standard back-off scoring (what KenLM computes for the
//                        same file; the reference's binary ".scorer" container is not read).
//
// Scoring follows pyctcdecode 0.x as published (parity unpinned: the package is not installed here):
// beam score = log-sum-exp of the CTC alignments of the prefix + sum over finished words of
// (alpha * ln10 * log10 p(word | history) [+ unk_offset for out-of-vocabulary words] + beta)
// + a penalty for an unfinished word that is not the prefix of any unigram (unk_offset, scaled by
// len/6 past six characters); per frame beams below best + beam_prune_logp are dropped and the best
// beam_width kept; at the end the open word is closed and the sentence-end probability added.
// One deliberate difference: a hypothesis ending in a space gets the sentence-end score of its text and
// nothing else (pyctcdecode scores an empty "word" there).
#include "common.h"

#include <algorithm>
#include <atomic>
#include <cstring>
#include <fstream>
#include <sstream>
#include <string>
#include <thread>
#include <unordered_map>
#include <vector>

namespace sp {

// ------------------------------------------------------------------------------------------------
// GPU: per-frame candidate lists
// ------------------------------------------------------------------------------------------------
constexpr float kLogMinP = -34.538776f;  // log(1e-15): pyctcdecode's MIN_TOKEN_CLIP_P

template <typename T>
__global__ __launch_bounds__(256) void beam_candidates_kernel(
    const T* __restrict__ scores, int rows, int Tn, int V, float min_logp, const int32_t* __restrict__ frame_lengths,
    int kmax, int32_t* __restrict__ cand_n, int32_t* __restrict__ cand_id, float* __restrict__ cand_lp) {
    const int lane = threadIdx.x & 63;
    const int row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const int b = row / Tn, t = row - b * Tn;
    if (frame_lengths && t >= frame_lengths[b]) {
        if (lane == 0) cand_n[row] = 0;
        return;
    }
    const T* x = scores + (int64_t)row * V;
    // pass 1: maximum and its first position
    float mx = -INFINITY;
    int arg = 0x7FFFFFFF;
    for (int c = lane; c < V; c += 64) {
        const float v = Elem<T>::ld(x + c);
        if (v > mx || arg == 0x7FFFFFFF) { mx = v; arg = c; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float om = __shfl_xor(mx, o);
        const int oa = __shfl_xor(arg, o);
        if (om > mx || (om == mx && oa < arg)) { mx = om; arg = oa; }
    }
    // pass 2: log-sum-exp
    float s = 0.f;
    for (int c = lane; c < V; c += 64) s += __expf(Elem<T>::ld(x + c) - mx);
    const float lse = mx + __logf(wave_sum(s));
    // pass 3: candidates in ascending id order
    int base = 0;
    int32_t* ids = cand_id + (int64_t)row * kmax;
    float* lps = cand_lp + (int64_t)row * kmax;
    for (int c0 = 0; c0 < V; c0 += 64) {
        const int c = c0 + lane;
        float lp = 0.f;
        bool keep = false;
        if (c < V) {
            lp = fminf(fmaxf(Elem<T>::ld(x + c) - lse, kLogMinP), 0.f);
            keep = lp >= min_logp || c == arg;
        }
        const unsigned long long m = __ballot(keep);
        const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
        if (keep && pos < kmax) { ids[pos] = c; lps[pos] = lp; }
        base += __popcll(m);
    }
    if (lane == 0) cand_n[row] = base;  // the true count: the host refuses a frame with more than kmax
}

// ------------------------------------------------------------------------------------------------
// n-gram model
// ------------------------------------------------------------------------------------------------
struct ProbBo { float prob, bo; };

static std::string ids_key(const int32_t* p, int n) { return std::string(reinterpret_cast<const char*>(p), n * 4); }

struct NgramLM {
    int order = 0;
    std::unordered_map<std::string, int32_t> vocab;
    std::vector<std::string> sorted_words;                       // for "is this a prefix of some unigram"
    std::vector<std::unordered_map<std::string, ProbBo>> tables; // tables[n-1]: n-grams keyed by their ids
    int32_t unk = -1, bos = -1, eos = -1;
    double alpha = 0.5, beta = 1.5, unk_offset = -10.0;
    bool score_boundary = true;

    int32_t word_id(const std::string& w) const {
        auto it = vocab.find(w);
        return it == vocab.end() ? -1 : it->second;
    }
    // log10 p(w | ctx) by back-off; out = the context after w (at most order-1 ids)
    double score(const std::vector<int32_t>& ctx, int32_t w, std::vector<int32_t>* out) const {
        const int32_t wq = w >= 0 ? w : unk;
        const int n = (int)ctx.size();
        double lp = 0.0;
        bool found = false;
        std::vector<int32_t> key(ctx);
        key.push_back(wq);
        if (wq >= 0) {
            for (int start = 0; start <= n; ++start) {
                const int len = n - start + 1;
                if (len <= order) {
                    auto it = tables[len - 1].find(ids_key(key.data() + start, len));
                    if (it != tables[len - 1].end()) { lp += it->second.prob; found = true; break; }
                    if (start < n) {
                        auto jt = tables[len - 2].find(ids_key(key.data() + start, len - 1));
                        if (jt != tables[len - 2].end()) lp += jt->second.bo;
                    }
                }
            }
        }
        if (!found) lp = -100.0;   // a model without <unk>: KenLM's unknown_missing_logprob
        if (out) {
            const int keep = std::min(n + 1, order - 1);
            out->assign(key.end() - keep, key.end());
        }
        return lp;
    }
    bool is_prefix_of_unigram(const std::string& part) const {
        auto it = std::lower_bound(sorted_words.begin(), sorted_words.end(), part);
        return it != sorted_words.end() && it->compare(0, part.size(), part) == 0;
    }
};

static int load_arpa(const char* path, NgramLM* lm) {
    std::ifstream in(path);
    if (!in) return fail(SP_EINVAL, "sp_ngram_load_arpa: cannot open %s", path);
    std::string line;
    int section = 0;   // 0 = header, n = inside \n-grams:
    std::vector<long> counts;
    while (std::getline(in, line)) {
        while (!line.empty() && (line.back() == '\r' || line.back() == ' ' || line.back() == '\t')) line.pop_back();
        if (line.empty()) continue;
        if (line[0] == '\\') {
            if (line == "\\data\\") { section = 0; continue; }
            if (line == "\\end\\") break;
            int n = 0;
            if (sscanf(line.c_str(), "\\%d-grams:", &n) == 1 && n >= 1) {
                section = n;
                if ((int)lm->tables.size() < n) lm->tables.resize(n);
                continue;
            }
            return fail(SP_EINVAL, "sp_ngram_load_arpa: unexpected line '%s'", line.c_str());
        }
        if (section == 0) {
            int n = 0; long c = 0;
            if (sscanf(line.c_str(), "ngram %d=%ld", &n, &c) == 2) { if ((int)counts.size() < n) counts.resize(n); counts[n - 1] = c; }
            continue;
        }
        std::istringstream ss(line);
        std::vector<std::string> tok;
        std::string t;
        while (ss >> t) tok.push_back(t);
        if ((int)tok.size() < section + 1 || (int)tok.size() > section + 2)
            return fail(SP_EINVAL, "sp_ngram_load_arpa: malformed %d-gram line '%s'", section, line.c_str());
        ProbBo pb{(float)atof(tok[0].c_str()), (int)tok.size() == section + 2 ? (float)atof(tok.back().c_str()) : 0.f};
        std::vector<int32_t> ids(section);
        for (int i = 0; i < section; ++i) {
            const std::string& w = tok[1 + i];
            auto it = lm->vocab.find(w);
            if (it == lm->vocab.end()) {
                if (section != 1) return fail(SP_EINVAL, "sp_ngram_load_arpa: '%s' of a %d-gram is not a unigram", w.c_str(), section);
                it = lm->vocab.emplace(w, (int32_t)lm->vocab.size()).first;
            }
            ids[i] = it->second;
        }
        lm->tables[section - 1][ids_key(ids.data(), section)] = pb;
    }
    lm->order = (int)lm->tables.size();
    if (lm->order < 1 || lm->vocab.empty()) return fail(SP_EINVAL, "sp_ngram_load_arpa: %s holds no n-grams", path);
    lm->unk = lm->word_id("<unk>");
    if (lm->unk < 0) lm->unk = lm->word_id("<UNK>");
    lm->bos = lm->word_id("<s>");
    lm->eos = lm->word_id("</s>");
    for (auto& kv : lm->vocab)
        if (kv.first != "<unk>" && kv.first != "<UNK>" && kv.first != "<s>" && kv.first != "</s>") lm->sorted_words.push_back(kv.first);
    std::sort(lm->sorted_words.begin(), lm->sorted_words.end());
    return SP_OK;
}

// ------------------------------------------------------------------------------------------------
// host: prefix beam search over candidate lists
// ------------------------------------------------------------------------------------------------
static void append_utf8(std::string& s, int32_t cp) {
    if (cp < 0x80) s.push_back((char)cp);
    else if (cp < 0x800) { s.push_back((char)(0xC0 | (cp >> 6))); s.push_back((char)(0x80 | (cp & 0x3F))); }
    else if (cp < 0x10000) {
        s.push_back((char)(0xE0 | (cp >> 12))); s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    } else {
        s.push_back((char)(0xF0 | (cp >> 18))); s.push_back((char)(0x80 | ((cp >> 12) & 0x3F)));
        s.push_back((char)(0x80 | ((cp >> 6) & 0x3F))); s.push_back((char)(0x80 | (cp & 0x3F)));
    }
}

static inline double log_add(double a, double b) {
    return a >= b ? a + std::log1p(std::exp(b - a)) : b + std::log1p(std::exp(a - b));
}

struct BeamParams {
    const int32_t* codepoints; int V, blank, space;
    const NgramLM* lm; int beam_width; double prune_logp; int n_best, max_out;
};

class Searcher {
public:
    explicit Searcher(const BeamParams& p) : P(p) {
        Node root;
        root.boundary = root_node();
        if (P.lm && P.lm->score_boundary && P.lm->bos >= 0) root.ctx.push_back(P.lm->bos);
        nodes.push_back(root);
    }
    // returns SP_OK or an error code
    int run(const int32_t* cand_n, const int32_t* cand_id, const float* cand_lp, int Tn, int kmax,
            int32_t* out_ids, int32_t* out_len, float* out_scores) {
        beams.assign(1, Beam{0, -1, 0.0, 0.0});
        for (int t = 0; t < Tn; ++t) {
            const int n = cand_n[t];
            if (n > kmax) return fail(SP_EINVAL, "sp_ctc_beam_decode: frame %d has %d candidates, the lists hold %d", t, n, kmax);
            if (n <= 0) return fail(SP_EINVAL, "sp_ctc_beam_decode: frame %d has no candidate", t);
            next.clear();
            slot.clear();
            for (int k = 0; k < n; ++k) {
                const int c = cand_id[(int64_t)t * kmax + k];
                if (c < 0 || c >= P.V) return fail(SP_EINVAL, "sp_ctc_beam_decode: candidate id %d outside [0,%d)", c, P.V);
                const double p = cand_lp[(int64_t)t * kmax + k];
                for (const Beam& bm : beams) {
                    int node = bm.node;
                    if (c == P.blank || c == bm.last) {
                        // blank, or a repeat of the token just emitted: the prefix stays
                    } else if (c == P.space) {
                        if (nodes[node].boundary != node) node = child(node, c);
                    } else {
                        node = child(node, c);
                    }
                    add(node, c, bm.logit + p);
                }
            }
            prune();
        }
        // close open words, merge hypotheses with the same text, add the sentence-end score
        next.clear();
        slot.clear();
        for (const Beam& bm : beams) {
            int node = bm.node;
            if (nodes[node].boundary != node) node = child(node, P.space);
            add(node, -1, bm.logit);
        }
        for (Beam& bm : next) {
            double lm = nodes[bm.node].lm_total;
            if (P.lm && P.lm->score_boundary && P.lm->eos >= 0)
                lm += P.lm->alpha * P.lm->score(nodes[bm.node].ctx, P.lm->eos, nullptr) * M_LN10;
            bm.score = bm.logit + lm;
        }
        finish_prune();
        for (int i = 0; i < P.n_best; ++i) {
            int32_t* o = out_ids + (int64_t)i * P.max_out;
            if (i >= (int)beams.size()) { out_len[i] = -1; out_scores[2 * i] = out_scores[2 * i + 1] = -INFINITY; continue; }
            std::vector<int32_t> lab;
            for (int nd = beams[i].node; nd != 0; nd = nodes[nd].parent) lab.push_back(nodes[nd].label);
            std::reverse(lab.begin(), lab.end());
            if (!lab.empty() && lab.back() == P.space) lab.pop_back();
            if ((int)lab.size() > P.max_out) return fail(SP_EINVAL, "sp_ctc_beam_decode: a hypothesis of %d tokens does not fit max_out %d", (int)lab.size(), P.max_out);
            std::copy(lab.begin(), lab.end(), o);
            std::fill(o + lab.size(), o + P.max_out, -1);
            out_len[i] = (int32_t)lab.size();
            out_scores[2 * i] = (float)beams[i].logit;
            out_scores[2 * i + 1] = (float)beams[i].score;
        }
        return SP_OK;
    }

private:
    struct Node {
        int parent = -1, label = -1;
        int boundary = 0;            // the node where the open word began (itself when no word is open)
        int wlen = 0;                // characters of the open word
        double lm_total = 0.0;       // boundary nodes: weighted LM score of the finished words
        double partial = 0.0;        // open word: penalty while it is the prefix of no unigram
        std::vector<int32_t> ctx;    // boundary nodes: LM context
    };
    struct Beam { int node, last; double logit, score; };
    static int root_node() { return 0; }

    std::string open_word(int node) const {
        std::vector<int32_t> cps;
        for (int nd = node; nd != nodes[node].boundary; nd = nodes[nd].parent) cps.push_back(P.codepoints[nodes[nd].label]);
        std::string w;
        for (auto it = cps.rbegin(); it != cps.rend(); ++it) append_utf8(w, *it);
        return w;
    }
    int child(int node, int c) {
        const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)c;
        auto it = children.find(key);
        if (it != children.end()) return it->second;
        Node nn;
        nn.parent = node;
        nn.label = c;
        const int id = (int)nodes.size();
        if (c == P.space) {              // closes the open word of `node`
            nn.boundary = id;
            const Node& start = nodes[nodes[node].boundary];
            nn.lm_total = start.lm_total;
            if (P.lm) {
                const std::string w = open_word(node);
                const int32_t wid = P.lm->word_id(w);
                double raw = P.lm->score(start.ctx, wid, &nn.ctx);
                if (wid < 0) raw += P.lm->unk_offset;
                nn.lm_total += P.lm->alpha * raw * M_LN10 + P.lm->beta;
            }
        } else {
            nn.boundary = nodes[node].boundary;
            nn.wlen = nodes[node].wlen + 1;
        }
        nodes.push_back(std::move(nn));
        if (c != P.space && P.lm) {
            Node& m = nodes.back();
            double pen = P.lm->is_prefix_of_unigram(open_word(id)) ? 0.0 : P.lm->unk_offset;
            if (m.wlen > 6) pen = pen * m.wlen / 6.0;
            m.partial = pen;
        }
        children.emplace(key, id);
        return id;
    }
    void add(int node, int last, double logit) {
        const uint64_t key = ((uint64_t)(uint32_t)node << 32) | (uint32_t)(last + 1);
        auto it = slot.find(key);
        if (it == slot.end()) {
            slot.emplace(key, (int)next.size());
            next.push_back(Beam{node, last, logit, 0.0});
        } else {
            Beam& b = next[it->second];
            b.logit = log_add(b.logit, logit);
        }
    }
    void prune() {
        for (Beam& bm : next) {
            const Node& nd = nodes[bm.node];
            bm.score = bm.logit + nodes[nd.boundary].lm_total + nd.partial;
        }
        finish_prune();
    }
    void finish_prune() {
        double best = -INFINITY;
        for (const Beam& bm : next) best = std::max(best, bm.score);
        beams.clear();
        for (const Beam& bm : next)
            if (bm.score >= best + P.prune_logp) beams.push_back(bm);
        std::stable_sort(beams.begin(), beams.end(), [](const Beam& a, const Beam& b) { return a.score > b.score; });
        if ((int)beams.size() > P.beam_width) beams.resize(P.beam_width);
    }

    const BeamParams& P;
    std::vector<Node> nodes;
    std::unordered_map<uint64_t, int> children, slot;
    std::vector<Beam> beams, next;
};

}  // namespace sp

extern "C" int sp_beam_candidates(const void* scores, int dtype, int B, int T, int V, float token_min_logp,
                                  const int32_t* frame_lengths, int kmax, int32_t* cand_n, int32_t* cand_id,
                                  float* cand_lp, void* stream) {
    using namespace sp;
    SP_CHECK_ARG(B >= 0 && T >= 0 && V >= 1 && kmax >= 1, "sp_beam_candidates: bad shape B=%d T=%d V=%d kmax=%d", B, T, V, kmax);
    SP_CHECK_ARG((int64_t)B * T < (1ll << 31), "sp_beam_candidates: B*T does not fit 31 bits");
    if (B == 0 || T == 0) return SP_OK;
    SP_CHECK_ARG(scores && cand_n && cand_id && cand_lp, "sp_beam_candidates: null pointer");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const int rows = B * T;
    dim3 grid((rows + 3) / 4), block(256);
    switch (dtype) {
        case SP_F32:
            hipLaunchKernelGGL(beam_candidates_kernel<float>, grid, block, 0, s, (const float*)scores, rows, T, V,
                               token_min_logp, frame_lengths, kmax, cand_n, cand_id, cand_lp);
            break;
        case SP_F16:
            hipLaunchKernelGGL(beam_candidates_kernel<__half>, grid, block, 0, s, (const __half*)scores, rows, T, V,
                               token_min_logp, frame_lengths, kmax, cand_n, cand_id, cand_lp);
            break;
        case SP_BF16:
            hipLaunchKernelGGL(beam_candidates_kernel<__hip_bfloat16>, grid, block, 0, s, (const __hip_bfloat16*)scores,
                               rows, T, V, token_min_logp, frame_lengths, kmax, cand_n, cand_id, cand_lp);
            break;
        default: return fail(SP_EINVAL, "sp_beam_candidates: unknown dtype %d", dtype);
    }
    return check_launch("beam_candidates_kernel");
}

extern "C" int sp_ngram_load_arpa(const char* path, float alpha, float beta, float unk_score_offset,
                                  int score_boundary, void** handle) {
    using namespace sp;
    SP_CHECK_ARG(path && handle, "sp_ngram_load_arpa: null pointer");
    NgramLM* lm = new NgramLM();
    lm->alpha = alpha; lm->beta = beta; lm->unk_offset = unk_score_offset; lm->score_boundary = score_boundary != 0;
    const int rc = load_arpa(path, lm);
    if (rc != SP_OK) { delete lm; return rc; }
    *handle = lm;
    return SP_OK;
}

extern "C" void sp_ngram_free(void* handle) { delete static_cast<sp::NgramLM*>(handle); }

extern "C" int sp_ngram_order(const void* handle) { return handle ? static_cast<const sp::NgramLM*>(handle)->order : 0; }

extern "C" int sp_ngram_score_sentence(const void* handle, const char* const* words, int n, int bos, int eos, double* log10_total) {
    using namespace sp;
    SP_CHECK_ARG(handle && log10_total && (n == 0 || words), "sp_ngram_score_sentence: null pointer");
    const NgramLM* lm = static_cast<const NgramLM*>(handle);
    std::vector<int32_t> ctx, nxt;
    if (bos && lm->bos >= 0) ctx.push_back(lm->bos);
    double total = 0.0;
    for (int i = 0; i < n; ++i) { total += lm->score(ctx, lm->word_id(words[i]), &nxt); ctx.swap(nxt); }
    if (eos && lm->eos >= 0) total += lm->score(ctx, lm->eos, nullptr);
    *log10_total = total;
    return SP_OK;
}

extern "C" int sp_ctc_beam_decode(const int32_t* cand_n, const int32_t* cand_id, const float* cand_lp, int B, int T,
                                  int kmax, const int32_t* frame_lengths, const int32_t* label_codepoints, int V,
                                  int blank, int space, const void* lm, int beam_width, float beam_prune_logp,
                                  int n_threads, int n_best, int max_out, int32_t* out_ids, int32_t* out_len,
                                  float* out_scores) {
    using namespace sp;
    SP_CHECK_ARG(B >= 0 && T >= 0 && kmax >= 1 && V >= 2, "sp_ctc_beam_decode: bad shape B=%d T=%d kmax=%d V=%d", B, T, kmax, V);
    SP_CHECK_ARG(blank >= 0 && blank < V && space >= -1 && space < V && space != blank,
                 "sp_ctc_beam_decode: blank %d / space %d outside [0,%d) or equal", blank, space, V);
    SP_CHECK_ARG(beam_width >= 1 && n_best >= 1 && max_out >= 0 && beam_prune_logp <= 0.f,
                 "sp_ctc_beam_decode: beam_width %d n_best %d max_out %d prune %g", beam_width, n_best, max_out, (double)beam_prune_logp);
    SP_CHECK_ARG(!lm || space >= 0, "sp_ctc_beam_decode: a language model needs a space token");
    if (B == 0) return SP_OK;
    SP_CHECK_ARG(cand_n && cand_id && cand_lp && label_codepoints && out_ids && out_len && out_scores, "sp_ctc_beam_decode: null pointer");
    BeamParams P{label_codepoints, V, blank, space, static_cast<const NgramLM*>(lm), beam_width, (double)beam_prune_logp, n_best, max_out};
    std::atomic<int> next_b{0};
    std::vector<int> rcs(B, SP_OK);
    std::vector<std::string> errs(B);
    auto work = [&]() {
        for (;;) {
            const int b = next_b.fetch_add(1);
            if (b >= B) break;
            const int Tb = frame_lengths ? std::min(std::max(frame_lengths[b], 0), T) : T;
            Searcher srch(P);
            rcs[b] = srch.run(cand_n + (int64_t)b * T, cand_id + (int64_t)b * T * kmax, cand_lp + (int64_t)b * T * kmax, Tb, kmax,
                              out_ids + (int64_t)b * n_best * max_out, out_len + (int64_t)b * n_best, out_scores + (int64_t)b * n_best * 2);
            if (rcs[b] != SP_OK) errs[b] = err_buf();   // the message is thread-local
        }
    };
    const int nt = std::max(1, std::min(n_threads, B));
    if (nt == 1) work();
    else {
        std::vector<std::thread> th;
        for (int i = 0; i < nt; ++i) th.emplace_back(work);
        for (auto& t : th) t.join();
    }
    for (int b = 0; b < B; ++b)
        if (rcs[b] != SP_OK) return fail(rcs[b], "%s (utterance %d)", errs[b].c_str(), b);
    return SP_OK;
}
