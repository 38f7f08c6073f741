// sbc_score_*: the whole NCSNv2Deepest score network behind ONE C call, for hosts that are not Python.
//
// score_based_channels_amd/plan.py + scorenet.py (the product's Python host) wire ncsnv2/models/ncsnv2.py:269-300 into ~150
// fused operator records, share activation storage between tensors with disjoint lifetimes, re-order the weights and bind
// everything to device memory.  This file does the same inside the library -- same fusion rules, same slot assignment, same
// packers -- so that a C / C++ / Go / Rust host gets a score evaluation from (state_dict tensors, batch size) without
// re-implementing any of it: sbc_score_create -> sbc_score_buffers -> sbc_score_forward, or sbc_score_ops to extend the
// record list with SBC_OP_LANGEVIN / SBC_OP_STEP_INC into a full annealed-Langevin step plan (sbc_plan_create).
// tests/test_gpu_capi.py holds it to the Python host: identical records, bit-identical outputs.
#include <string.h>
#include <map>
#include <string>
#include <vector>
#include "common.h"

namespace {

using namespace sbc;

struct Tn { std::string name; int h, w, c; int slot = -1; };

struct POp {
    int kind = 0, flags = 0, ksize = 3, dil = 1, tag = 0;
    int src = -1, dst = -1, stats = -1, res1 = -1, res2 = -1, up = -1;       // tensor indices
    int moments = -1;                // second output: tile moments of dst (SBC_EPI_MOMENTS_OUT)
    int geom = -1;                   // INORM_STATS from tile moments: the tensor whose (H, W, C) the launch describes
    std::string norm_key;            // CONV with SBC_PRO_NORM_SELF: the norm whose (alpha | gamma | beta) `stats` points at (plan.py)
    std::string weight, bias, weight2;       // weight2: the second convolution of an SBC_OP_CONV_PAIR / SBC_OP_RES_BLOCK
    std::string bias2, norm2;                // SBC_OP_RES_BLOCK: the second convolution's bias, the second norm (plan.py)
    struct Block {                           // plan.py: Op.blocks; RES blocks carry the rest
        int type; std::string w1, w2;
        int dil = 1; std::string bias1, bias2, norm1, norm2, w3, bias3;
    };
    std::vector<Block> blocks;               // SBC_OP_CHAIN: the RCU / CRP blocks in execution order (plan.py: Op.blocks)
    int lane = 0, signal = 0, wait0 = 0, wait1 = 0;   // launch lanes (sbc_op.lane / signal / wait; plan.hoist_skip_branches)
    // the state_dict key that names the record's layer (plan.py: Op.name has the same prefix)
    const std::string& label() const { return !blocks.empty() ? blocks[0].w1 : !weight.empty() ? weight : norm_key; }
};

// ---- wiring: a transcription of plan.py's _Builder (reference lines cited there) --------------------------------
struct Builder {
    int ngf, nt, nr;
    bool fuse_pairs = false;         // plan.py: fuse_pairs / pair_fusable
    bool fuse_res = false;           // plan.py: fuse_res / res_fusable (SBC_OP_RES_BLOCK; conv_mode f16x2 with fused pairs)
    bool f16w = false;               // ... the fp16-weight mode also fuses 64-pixel rows (plan.PAIR_WIDTHS_F16W)
    bool fold_stats = false;         // plan.py: fold_stats (statistics of full-resolution tensors from their producers' tile moments)
    bool fuse_down = false;          // plan.py: fuse_down / down_fusable (SBC_OP_CONV_DOWN; conv_mode f16x2)
    bool fuse_chain = false;         // plan.py: fuse_chain / chain_fusable (SBC_OP_CHAIN at the 8 x 2 level; conv_mode f16x2)
    std::map<int, int> producer;     // tensor -> index of the record that writes it
    std::vector<Tn> t;
    std::vector<POp> ops;
    int tensor(const std::string& n, int h, int w, int c) { t.push_back({n, h, w, c}); return (int)t.size() - 1; }
    int conv(const std::string& name, int src, const std::string& wkey, int cout, bool bias = true, int flags = 0,
             int stats = -1, int res1 = -1, int res2 = -1, int up = -1, int ksize = 3, int dil = 1) {
        const bool pool = flags & SBC_EPI_POOL;
        const int sh = t[src].h, sw = t[src].w, sc = t[src].c;
        const int dst = tensor(name, pool ? sh / 2 : sh, pool ? sw / 2 : sw, cout);
        POp o;
        o.kind = SBC_OP_CONV; o.src = src; o.dst = dst; o.weight = wkey + ".weight";
        if (bias) o.bias = wkey + ".bias";
        o.stats = stats; o.res1 = res1; o.res2 = res2; o.up = up;
        o.flags = flags | (up >= 0 ? SBC_EPI_UP : 0); o.ksize = ksize; o.dil = dil;
        if (stats == SELF_NORM) { o.stats = -1; o.norm_key = pending_norm; o.flags |= SBC_PRO_NORM_SELF; }   // plan.py: SelfNorm
        producer[dst] = (int)ops.size();
        o.tag = (ksize == 3 && sc == ngf && cout == ngf && sh == nt) ? 1 : 0;                 // plan.TAG_CONV_TOP
        if (ksize == 3 && dil == 1 && sc == 2 * ngf && cout == 2 * ngf && !pool && 2 * sh == nt) o.tag = 3;   // plan.TAG_CONV_MID
        ops.push_back(o);
        return dst;
    }
    static constexpr int SELF_NORM = -2;   // stats(): no statistics record, the consuming convolution computes them (plan.py: SelfNorm)
    std::string pending_norm;
    int stats(const std::string& name, int src, const std::string& nkey, bool consumer_is_conv = true) {
        const int hw = t[src].h * t[src].w, sw = t[src].w;
        if (fold_stats && consumer_is_conv && hw <= 64 && !(hw & (hw - 1)) && sw >= 2 && !(sw & (sw - 1))) {
            pending_norm = nkey;
            return SELF_NORM;
        }
        const int dst = tensor(name, 1, 3, t[src].c);
        POp o; o.kind = SBC_OP_INORM_STATS; o.src = src; o.dst = dst; o.weight = nkey;
        auto it = producer.find(src);
        if (fold_stats && it != producer.end() && (t[src].c == 32 || t[src].c == 64) && hw % 128 == 0 && hw >= 256 &&
            128 % (2 * sw) == 0 && t[src].h % (128 / sw > 0 ? 128 / sw : 1) == 0) {
            POp& pr = ops[it->second];
            if ((pr.kind == SBC_OP_BEGIN_CONV && t[src].c == 32) ||
                (pr.kind == SBC_OP_CONV && pr.ksize == 3 && pr.dil == 1 && !(pr.flags & SBC_EPI_POOL)) || pr.kind == SBC_OP_RES_BLOCK) {
                if (pr.moments < 0) {
                    pr.moments = tensor(t[src].name + ".moments", hw / 128, t[src].c, 2);
                    pr.flags |= SBC_EPI_MOMENTS_OUT;
                }
                o.src = ops[it->second].moments; o.flags = SBC_PRO_NORM_MOMENTS; o.geom = src;
            }
        }
        ops.push_back(o);
        return dst;
    }
    int maxpool(const std::string& name, int src, bool elu) {
        const int dst = tensor(name, t[src].h, t[src].w, t[src].c);
        POp o; o.kind = SBC_OP_MAXPOOL5; o.src = src; o.dst = dst; o.flags = elu ? SBC_PRO_ELU : 0;
        ops.push_back(o);
        return dst;
    }
    int residual_block(const std::string& p, int x, int cout, bool down, int dilation) {      // layers.py:443-456
        const int d = dilation ? dilation : 1;
        const bool pooled = down && !dilation;
        const int c1 = down ? t[x].c : cout;
        if (chain_fusable(x, SBC_CHAIN_RES) && t[x].c == cout && !pooled && (d == 1 || t[x].w == 2)) {       // plan.py: a RES block of a CHAIN record
            POp::Block bl{SBC_CHAIN_RES, p + "conv1.weight", p + "conv2.weight"};
            bl.dil = d; bl.bias1 = p + "conv1.bias"; bl.bias2 = p + "conv2.bias"; bl.norm1 = p + "normalize1"; bl.norm2 = p + "normalize2";
            if (down) { bl.w3 = p + "shortcut.weight"; bl.bias3 = p + "shortcut.bias"; }
            return chain(p + "chain", x, {bl});
        }
        const int s1 = stats(p + "normalize1", x, p + "normalize1");
        if (fuse_res && t[x].c == 32 && cout == 32 && !down && !dilation && t[x].h == 64 && t[x].w == 16 && s1 != SELF_NORM) {   // plan.res_fusable
            const int out = tensor(p + "conv2", t[x].h, t[x].w, cout);
            POp o;
            o.kind = SBC_OP_RES_BLOCK; o.src = x; o.dst = out; o.stats = s1;
            o.weight = p + "conv1.weight"; o.weight2 = p + "conv2.weight"; o.bias = p + "conv1.bias"; o.bias2 = p + "conv2.bias";
            o.norm2 = p + "normalize2";
            o.tag = t[x].h == nt ? 6 : 0;                                                       // plan.TAG_RES_TOP
            producer[out] = (int)ops.size();
            ops.push_back(o);
            return out;
        }
        const int a = conv(p + "conv1", x, p + "conv1", c1, true, SBC_PRO_NORM | SBC_PRO_ELU, s1, -1, -1, -1, 3, d);
        const int s2 = stats(p + "normalize2", a, p + "normalize2");
        if (pooled && fuse_down && s2 != SELF_NORM && t[x].h % 16 == 0 &&
            ((t[x].w == 16 && t[x].c == 32 && cout == 64) || (t[x].w == 8 && t[x].c == 64 && cout == 64))) {        // plan.down_fusable
            const int out = tensor(p + "conv2", t[x].h / 2, t[x].w / 2, cout);
            POp o;
            o.kind = SBC_OP_CONV_DOWN; o.src = a; o.dst = out; o.stats = s2; o.res1 = x;
            o.weight = p + "conv2.conv.weight"; o.weight2 = p + "shortcut.conv.weight";
            o.bias = p + "conv2.conv.bias"; o.bias2 = p + "shortcut.conv.bias";
            o.tag = 12 + (t[x].w == 16 ? 0 : 1);                                                // plan.TAG_DOWN
            producer[out] = (int)ops.size();
            ops.push_back(o);
            return out;
        }
        if (pooled) {
            const int sc = conv(p + "shortcut", x, p + "shortcut.conv", cout, true, SBC_EPI_POOL, -1, -1, -1, -1, 1, 1);
            return conv(p + "conv2", a, p + "conv2.conv", cout, true, SBC_PRO_NORM | SBC_PRO_ELU | SBC_EPI_POOL, s2, sc);
        }
        int sc = x;
        if (t[x].c != cout || down) sc = conv(p + "shortcut", x, p + "shortcut", cout, true, 0, -1, -1, -1, -1, 3, d);
        return conv(p + "conv2", a, p + "conv2", cout, true, SBC_PRO_NORM | SBC_PRO_ELU, s2, sc, -1, -1, 3, d);
    }
    bool chain_fusable(int x, int kind = SBC_CHAIN_RCU) const {                                // plan.chain_fusable
        if (!fuse_chain) return false;
        if (t[x].h == 32 && t[x].w == 8 && (t[x].c == 32 || t[x].c == 64)) return kind != SBC_CHAIN_RES;
        return (t[x].h == 8 && t[x].w == 2 && (t[x].c == 64 || t[x].c == 128)) || (t[x].h == 16 && t[x].w == 4 && t[x].c == 64);
    }
    static std::vector<POp::Block> rcu_blocks(const std::string& p, int n_blocks) {             // plan._Builder.rcu_blocks
        std::vector<POp::Block> v;
        for (int i = 1; i <= n_blocks; ++i)
            v.push_back(POp::Block{SBC_CHAIN_RCU, p + std::to_string(i) + "_1_conv.weight", p + std::to_string(i) + "_2_conv.weight"});
        return v;
    }
    int chain(const std::string& name, int x, const std::vector<POp::Block>& blocks) {          // plan._Builder.chain
        for (size_t k = 0; k < blocks.size(); k += SBC_CHAIN_MAX_BLOCKS) {
            const int dst = tensor(name + "." + std::to_string(k / SBC_CHAIN_MAX_BLOCKS), t[x].h, t[x].w, t[x].c);
            POp o;
            o.kind = SBC_OP_CHAIN; o.src = x; o.dst = dst;
            // plan.TAG_CHAIN + index in plan.CHAIN_KERNELS: (128, 2), (64, 2), (64, 4), (64, 8), (32, 8)
            o.tag = 7 + (t[x].c == 128 ? 0 : t[x].w == 2 ? 1 : t[x].w == 4 ? 2 : t[x].c == 64 ? 3 : 4);
            o.blocks.assign(blocks.begin() + k, blocks.begin() + std::min(blocks.size(), k + (size_t)SBC_CHAIN_MAX_BLOCKS));
            producer[dst] = (int)ops.size();
            ops.push_back(o);
            x = dst;
        }
        return x;
    }
    int rcu(const std::string& p, int x, int n_blocks) {                                        // layers.py:126-134
        if (chain_fusable(x)) return chain(p + "chain", x, rcu_blocks(p, n_blocks));
        for (int i = 1; i <= n_blocks; ++i) {
            const std::string a = p + std::to_string(i) + "_1_conv", b = p + std::to_string(i) + "_2_conv";
            const int pc = t[x].c, pw = t[x].w, ph = t[x].h;                                // plan.pair_fusable / PAIR_SHAPES*
            if (fuse_pairs && ((pc == 32 && pw == 16 && ph % 8 == 0) ||
                               (f16w && ((pc == 32 && (pw == 32 || pw == 64) && ph % 4 == 0) || (pc == 64 && pw == 16 && ph % 8 == 0) ||
                                         (pc == 64 && pw == 32 && ph % 4 == 0))))) {
                const int dst = tensor(b, t[x].h, t[x].w, t[x].c);
                POp o;
                o.kind = SBC_OP_CONV_PAIR; o.src = x; o.dst = dst; o.weight = a + ".weight"; o.weight2 = b + ".weight";
                o.tag = t[x].h == nt ? 2 : 0;                                               // plan.TAG_PAIR_TOP
                ops.push_back(o);
                x = dst;
                continue;
            }
            const int tt = conv(a, x, a, t[x].c, false, SBC_PRO_ELU);
            x = conv(b, tt, b, t[x].c, false, SBC_PRO_ELU, -1, x);
        }
        return x;
    }
    int crp(const std::string& p, int x) {                                                     // layers.py:76-83
        if (fuse_pairs && t[x].c == 32 && t[x].w == 16 && t[x].h % 8 == 0) {                   // plan.pool_fusable: SBC_OP_CONV_POOL
            const int path0 = tensor(p + "convs.0", t[x].h, t[x].w, t[x].c);
            POp a;
            a.kind = SBC_OP_CONV_POOL; a.src = x; a.dst = path0; a.weight = p + "convs.0.weight"; a.flags = SBC_PRO_ELU;
            a.tag = t[x].h == nt ? 4 : 0;                                                       // plan.TAG_POOL_TOP
            producer[path0] = (int)ops.size();
            ops.push_back(a);
            const int out = tensor(p + "convs.1", t[x].h, t[x].w, t[x].c);
            POp b;
            b.kind = SBC_OP_CONV_POOL; b.src = path0; b.dst = out; b.weight = p + "convs.1.weight"; b.flags = SBC_EPI_RES1_ELU;
            b.res1 = x; b.res2 = path0; b.tag = a.tag;
            producer[out] = (int)ops.size();
            ops.push_back(b);
            return out;
        }
        const int p0 = maxpool(p + "pool0", x, true);
        const int path0 = conv(p + "convs.0", p0, p + "convs.0", t[x].c, false);
        const int p1 = maxpool(p + "pool1", path0, false);
        return conv(p + "convs.1", p1, p + "convs.1", t[x].c, false, SBC_EPI_RES1_ELU, -1, x, path0);
    }
    // layers.py:234-249 with MSF (layers.py:178-184): the second input's adapt convolutions and its MSF convolution come
    // first (they do not depend on the first input's; plan.py issues them as side records when asked to overlap)
    int refine(const std::string& p, const std::vector<int>& xs, int features, bool end = false) {
        int h;
        const POp::Block crp_block{SBC_CHAIN_CRP, p + "crp.convs.0.weight", p + "crp.convs.1.weight"};
        if (xs.size() == 1 && chain_fusable(xs[0], SBC_CHAIN_CRP) && features == t[xs[0]].c) {              // the whole RefineBlock is one chain
            std::vector<POp::Block> bl = rcu_blocks(p + "adapt_convs.0.", 2);
            bl.push_back(crp_block);
            for (const auto& r : rcu_blocks(p + "output_convs.", end ? 3 : 1)) bl.push_back(r);
            return chain(p + "chain", xs[0], bl);
        }
        if (xs.size() == 1) {
            h = rcu(p + "adapt_convs.0.", xs[0], 2);
        } else {
            const int h1 = rcu(p + "adapt_convs.1.", xs[1], 2);
            const int t1 = conv(p + "msf.convs.1", h1, p + "msf.convs.1", features);
            const int h0 = rcu(p + "adapt_convs.0.", xs[0], 2);
            h = conv(p + "msf.convs.0", h0, p + "msf.convs.0", features, true, 0, -1, -1, -1, t1);
        }
        if (chain_fusable(h, SBC_CHAIN_CRP)) {
            std::vector<POp::Block> bl{crp_block};
            for (const auto& r : rcu_blocks(p + "output_convs.", end ? 3 : 1)) bl.push_back(r);
            return chain(p + "tail", h, bl);
        }
        h = crp(p + "crp.", h);
        return rcu(p + "output_convs.", h, end ? 3 : 1);
    }
};

}  // namespace

struct sbc_score {
    sbc_score_desc desc;
    std::vector<Tn> tensors;
    std::vector<POp> pops;
    std::vector<size_t> slot_elems;
    std::vector<float*> slots;           // device, [B * slot_elems]
    float* wdev = nullptr;               // all parameters, packed
    float* sigmas = nullptr;             // device [num_classes]
    int64_t* labels = nullptr;           // device [B]
    sbc_endconv endc;
    std::vector<sbc_op> ops;
    std::vector<sbc_chain> chains;       // ext structs of the SBC_OP_CHAIN records (sized before the records point into it)
    sbc_plan* plan = nullptr;
    int x_t = -1, out_t = -1;
};

namespace {

void assign_slots(sbc_score& s) {       // plan.assign_slots: linear scan, outputs never alias inputs of their own op
    const int n_ops = (int)s.pops.size();
    std::vector<int> last_use(s.tensors.size(), -1);
    // a record on a lane is known to be complete at the first run-stream record that waits for an event its lane signals at or behind it
    // (plan.assign_slots); without one, at the end of the list
    std::vector<int> done_at(n_ops);
    for (int i = 0; i < n_ops; ++i) {
        done_at[i] = i;
        if (!s.pops[i].lane) continue;
        done_at[i] = n_ops - 1;
        for (int k = i; k < n_ops && done_at[i] == n_ops - 1; ++k) {
            if (s.pops[k].lane != s.pops[i].lane || !s.pops[k].signal) continue;
            for (int m = k + 1; m < n_ops; ++m)
                if (!s.pops[m].lane && (s.pops[m].wait0 == s.pops[k].signal || s.pops[m].wait1 == s.pops[k].signal)) { done_at[i] = m; break; }
        }
    }
    for (int i = 0; i < n_ops; ++i)
        for (int id : {s.pops[i].src, s.pops[i].stats, s.pops[i].res1, s.pops[i].res2, s.pops[i].up})
            if (id >= 0 && done_at[i] > last_use[id]) last_use[id] = done_at[i];
    std::map<size_t, std::vector<int>> free_slots;
    std::vector<int> live;
    auto elems = [&](int id) { return (size_t)s.tensors[id].h * s.tensors[id].w * s.tensors[id].c; };
    auto pinned = [&](int id) { return id == s.x_t || id == s.out_t; };
    auto alloc = [&](int id) {
        auto& pool = free_slots[elems(id)];
        if (!pool.empty() && !pinned(id)) { s.tensors[id].slot = pool.back(); pool.pop_back(); }
        else { s.tensors[id].slot = (int)s.slot_elems.size(); s.slot_elems.push_back(elems(id)); }
    };
    alloc(s.x_t);
    for (int i = 0; i < n_ops; ++i) {
        const int dst = s.pops[i].dst, mom = s.pops[i].moments;
        alloc(dst);
        live.push_back(dst);
        if (mom >= 0) { alloc(mom); live.push_back(mom); }
        for (size_t k = 0; k < live.size();) {
            const int id = live[k];
            if (!pinned(id) && id != dst && id != mom && last_use[id] <= i) {
                free_slots[elems(id)].push_back(s.tensors[id].slot);
                live.erase(live.begin() + k);
            } else {
                ++k;
            }
        }
    }
}

float round_f16(float f) { return (float)(_Float16)f; }

}  // namespace

extern "C" {

int sbc_score_create(const sbc_score_desc* d, const sbc_tensor_ref* tensors, int32_t n_tensors, sbc_score** out) {
    SBC_REQUIRE(d && tensors && out && n_tensors > 0, "sbc_score_create: bad arguments");
    SBC_REQUIRE(d->ngf == 32 && d->channels == 2, "sbc_score_create: kernels are instantiated for ngf = 32, 2 channels");
    SBC_REQUIRE(d->nt > 0 && d->nr > 0 && d->nt % 8 == 0 && d->nr % 8 == 0,
                "sbc_score_create: Nt and Nr must be multiples of 8 (three 2x mean pools), got %dx%d", d->nt, d->nr);
    SBC_REQUIRE(d->batch > 0 && d->conv_mode >= 0 && d->conv_mode <= 3 && d->sigmas && d->num_classes > 0,
                "sbc_score_create: batch, conv_mode in {0 bf16x3, 1 f32, 2 f16w, 3 f16x2}, sigmas required");
    SBC_REQUIRE(!(d->flags & SBC_SCORE_FUSE_RES) || d->conv_mode == 3, "sbc_score_create: SBC_SCORE_FUSE_RES needs conv_mode 3 (f16x2)");
    SBC_REQUIRE(!(d->flags & SBC_SCORE_FUSE_CHAIN) || d->conv_mode == 3, "sbc_score_create: SBC_SCORE_FUSE_CHAIN needs conv_mode 3 (f16x2)");
    SBC_REQUIRE(!(d->flags & SBC_SCORE_FUSE_DOWN) || d->conv_mode == 3, "sbc_score_create: SBC_SCORE_FUSE_DOWN needs conv_mode 3 (f16x2)");
    SBC_REQUIRE(!(d->flags & SBC_SCORE_FUSE_PAIRS) || d->conv_mode >= 2,
                "sbc_score_create: SBC_SCORE_FUSE_PAIRS needs the fp16 weight forms (conv_mode 2 or 3)");
    std::map<std::string, const sbc_tensor_ref*> sd;
    for (int i = 0; i < n_tensors; ++i) {
        SBC_REQUIRE(tensors[i].name && tensors[i].data, "sbc_score_create: tensor %d has no name / data", i);
        sd[tensors[i].name] = &tensors[i];
    }
    sbc_score* s = new sbc_score();
    s->desc = *d;
    const int ngf = d->ngf, nt = d->nt, nr = d->nr, B = d->batch;
    // ---- wiring (plan.build_score_plan)
    Builder b{ngf, nt, nr};
    b.fuse_pairs = (d->flags & SBC_SCORE_FUSE_PAIRS) != 0;
    b.f16w = d->conv_mode == 2;
    b.fuse_res = (d->flags & SBC_SCORE_FUSE_RES) != 0;
    b.fuse_chain = (d->flags & SBC_SCORE_FUSE_CHAIN) != 0;
    b.fuse_down = (d->flags & SBC_SCORE_FUSE_DOWN) != 0;
    b.fold_stats = (d->flags & SBC_SCORE_FOLD_STATS) != 0 && d->conv_mode != 1 && !(nt & (nt - 1)) && !(nr & (nr - 1));
    const int x = b.tensor("x", nt, nr, d->channels);
    int h = b.tensor("begin_conv", nt, nr, ngf);
    { POp o; o.kind = SBC_OP_BEGIN_CONV; o.src = x; o.dst = h; o.weight = "begin_conv.weight"; o.bias = "begin_conv.bias"; b.producer[h] = (int)b.ops.size(); b.ops.push_back(o); }
    struct Stage { const char* name; int cout; bool down; int dil; };
    const Stage stages[6] = {{"res1", ngf, false, 0}, {"res2", 2 * ngf, true, 0}, {"res3", 2 * ngf, true, 0},
                             {"res31", 2 * ngf, true, 0}, {"res4", 4 * ngf, true, 2}, {"res5", 4 * ngf, true, 4}};
    int layers[6];
    for (int i = 0; i < 6; ++i) {
        h = b.residual_block(std::string(stages[i].name) + ".0.", h, stages[i].cout, stages[i].down, stages[i].dil);
        h = b.residual_block(std::string(stages[i].name) + ".1.", h, stages[i].cout, false, stages[i].dil);
        layers[i] = h;
    }
    const int ref1 = b.refine("refine1.", {layers[5]}, 4 * ngf);
    const int ref2 = b.refine("refine2.", {layers[4], ref1}, 2 * ngf);
    const int ref31 = b.refine("refine31.", {layers[3], ref2}, 2 * ngf);
    const int ref3 = b.refine("refine3.", {layers[2], ref31}, 2 * ngf);
    const int ref4 = b.refine("refine4.", {layers[1], ref3}, ngf);
    const int ref5 = b.refine("refine5.", {layers[0], ref4}, ngf, true);
    const int o_t = b.tensor("score", nt, nr, d->channels);
    if ((d->flags & SBC_SCORE_FUSE_END) && ngf == 32 && nt * nr == 1024) {       // plan.end_fusable
        POp o; o.kind = SBC_OP_END_CONV; o.src = ref5; o.dst = o_t; o.weight = "end_conv.weight"; o.bias = "end_conv.bias";
        o.flags = SBC_PRO_NORM_SELF; o.norm_key = "normalizer"; b.ops.push_back(o);
    } else {
        const int sn = b.stats("normalizer", ref5, "normalizer", false);
        POp o; o.kind = SBC_OP_END_CONV; o.src = ref5; o.dst = o_t; o.weight = "end_conv.weight"; o.bias = "end_conv.bias"; o.stats = sn; b.ops.push_back(o);
    }
    // plan.merge_chains: adjacent CHAIN records, the second the only consumer of the first one's output, become one record
    for (size_t k = 0; k + 1 < b.ops.size();) {
        POp& a = b.ops[k];
        const POp& c = b.ops[k + 1];
        bool ok = a.kind == SBC_OP_CHAIN && c.kind == SBC_OP_CHAIN && c.src == a.dst && a.blocks.size() + c.blocks.size() <= SBC_CHAIN_MAX_BLOCKS;
        for (size_t j = k + 2; ok && j < b.ops.size(); ++j)
            for (int id : {b.ops[j].src, b.ops[j].stats, b.ops[j].res1, b.ops[j].res2, b.ops[j].up}) ok = ok && id != a.dst;
        if (ok) {
            a.blocks.insert(a.blocks.end(), c.blocks.begin(), c.blocks.end());
            a.dst = c.dst;
            b.ops.erase(b.ops.begin() + k + 1);
        } else {
            ++k;
        }
    }
    for (auto& o : b.ops)                   // plan.TAG_DIRECT_MID
        if (o.tag == 3 && !(o.flags & (SBC_PRO_NORM | SBC_EPI_UP | SBC_EPI_MOMENTS_OUT))) o.tag = 5;
    if (d->flags & SBC_SCORE_SKIP_LANES) {
        // plan.hoist_skip_branches with plan.DEFAULT_SKIP_SPEC: the decoder's skip branches behind their anchors, on lane 1
        static const char* const spec[5][2] = {{"refine5.adapt_convs.0.", "res3.1."}, {"refine4.adapt_convs.0.", "res3.1."}, {"refine3.adapt_convs.0.", "res3.1."},
                                               {"refine31.adapt_convs.0.", "res31.1."}, {"refine2.adapt_convs.0.", "res4.1."}};
        auto starts = [](const std::string& v, const char* pre) { return v.compare(0, strlen(pre), pre) == 0; };
        int next_evt = 1;
        auto signal_of = [&](POp& o) { if (!o.signal) o.signal = next_evt++; return o.signal; };
        for (const auto& e : spec) {
            int i0 = -1, i1 = -1;
            for (int i = 0; i < (int)b.ops.size(); ++i)
                if (starts(b.ops[i].label(), e[0])) { if (i0 < 0) i0 = i; i1 = i; }
            if (i0 < 0) continue;                                   // (a plan without this branch as records of its own)
            for (int i = i0; i <= i1; ++i) SBC_REQUIRE(starts(b.ops[i].label(), e[0]) && !b.ops[i].lane, "sbc_score_create: skip branch %s is not one run of records", e[0]);
            int a = -1;
            for (int i = 0; i < i0; ++i) if (starts(b.ops[i].label(), e[1])) a = i;
            SBC_REQUIRE(a >= 0, "sbc_score_create: anchor %s of skip branch %s not found", e[1], e[0]);
            std::vector<POp> branch(b.ops.begin() + i0, b.ops.begin() + i1 + 1);
            const int result = branch.back().dst;
            b.ops.erase(b.ops.begin() + i0, b.ops.begin() + i1 + 1);
            int consumer = -1;
            for (int i = i0; i < (int)b.ops.size() && consumer < 0; ++i)
                for (int id : {b.ops[i].src, b.ops[i].stats, b.ops[i].res1, b.ops[i].res2, b.ops[i].up}) if (id == result) consumer = i;
            SBC_REQUIRE(consumer >= 0 && !b.ops[consumer].wait1, "sbc_score_create: skip branch %s has no consumer with a free wait slot", e[0]);
            for (auto& o : branch) o.lane = 1;
            branch.front().wait0 = signal_of(b.ops[a]);
            const int done = signal_of(branch.back());
            (b.ops[consumer].wait0 ? b.ops[consumer].wait1 : b.ops[consumer].wait0) = done;
            b.ops.insert(b.ops.begin() + a + 1, branch.begin(), branch.end());
        }
    }
    s->tensors = b.t; s->pops = b.ops; s->x_t = x; s->out_t = o_t;
    assign_slots(*s);

    // ---- parameters: one flat host image (16-byte aligned entries), packed by the library's own packers
    const bool f16w = d->conv_mode == 2, f16x2 = d->conv_mode == 3;
    std::vector<float> host;
    std::map<std::string, size_t> off;
    auto reserve = [&](const std::string& key, size_t n) {
        host.resize((host.size() + 3) / 4 * 4);
        off[key] = host.size();
        host.resize(host.size() + n);
        return host.data() + off[key];
    };
    auto find = [&](const std::string& key, int64_t numel) -> const float* {
        auto it = sd.find(key);
        if (it == sd.end()) { set_error("sbc_score_create: tensor '%s' missing from the state dict", key.c_str()); return nullptr; }
        if (it->second->numel != numel) {
            set_error("sbc_score_create: tensor '%s' has %lld elements, expected %lld", key.c_str(),
                      (long long)it->second->numel, (long long)numel);
            return nullptr;
        }
        return it->second->data;
    };
    auto fail = [&]() { sbc_score_destroy(s); return SBC_ERR_INVALID; };
    std::vector<float> tmp;
    auto rounded = [&](const float* p, size_t n) {      // fp16 parameters for conv_mode f16w (module.half() semantics)
        if (!f16w) return p;
        tmp.assign(p, p + n);
        for (auto& v : tmp) v = round_f16(v);
        return (const float*)tmp.data();
    };
    for (const POp& o : s->pops) {
        const Tn& src = s->tensors[o.geom >= 0 ? o.geom : o.src];
        const Tn& dst = s->tensors[o.dst];
        const std::string& nkey = o.kind == SBC_OP_INORM_STATS ? o.weight : o.kind == SBC_OP_RES_BLOCK ? o.norm2 : o.norm_key;
        if (!nkey.empty() && !off.count(nkey)) {             // a norm's (alpha | gamma | beta), [3][C]
            for (int k = 0; k < 3; ++k) {
                const char* suffix[3] = {".alpha", ".gamma", ".beta"};
                const float* v = find(nkey + suffix[k], src.c);
                if (!v) return fail();
                if (k == 0) reserve(nkey, 3 * (size_t)src.c);
                v = rounded(v, src.c);
                memcpy(host.data() + off[nkey] + (size_t)k * src.c, v, sizeof(float) * src.c);
            }
        }
        for (const auto& bl : o.blocks)                      // the norms of the RES blocks of a CHAIN record
            for (const std::string& nk : {bl.norm1, bl.norm2}) {
                if (nk.empty() || off.count(nk)) continue;
                for (int k = 0; k < 3; ++k) {
                    const char* suffix[3] = {".alpha", ".gamma", ".beta"};
                    const float* v = find(nk + suffix[k], src.c);
                    if (!v) return fail();
                    if (k == 0) reserve(nk, 3 * (size_t)src.c);
                    v = rounded(v, src.c);
                    memcpy(host.data() + off[nk] + (size_t)k * src.c, v, sizeof(float) * src.c);
                }
            }
        if (o.kind == SBC_OP_INORM_STATS) continue;
        if (o.weight.empty() && o.blocks.empty()) continue;   // max pooling has no parameters
        const int k = o.ksize, cin = src.c, cout = dst.c;
        const size_t wn = (size_t)cout * cin * k * k;
        std::vector<std::string> wkeys{o.weight, o.weight2};
        for (const auto& bl : o.blocks) { wkeys.push_back(bl.w1); wkeys.push_back(bl.w2); wkeys.push_back(bl.w3); }
        for (const std::string& wkey : wkeys) {
        if (wkey.empty()) continue;
        if (o.kind == SBC_OP_CONV_DOWN) {
            // the pooled stride-2 filters: 3x3 -> 4x4 (conv2), 1x1 -> 2x2 (shortcut)   (scorenet.load_state_dict: '#pool')
            if (off.count(wkey + "#pool")) continue;
            const int kk = wkey == o.weight2 ? 1 : 3;
            const float* w2 = find(wkey, (int64_t)cout * cin * kk * kk);
            if (!w2) return fail();
            sbc_pack_conv_weight_pooled_f16x2(w2, cout, cin, kk, (uint16_t*)reserve(wkey + "#pool", sbc_f16x2_elems((kk + 1) * (kk + 1), cin, cout) / 2));
            continue;
        }
        if (!off.count(wkey) && !off.count(wkey + "#split")) {
            const float* w = find(wkey, (int64_t)wn);
            if (!w) return fail();
            w = rounded(w, wn);
            std::vector<float> wkeep(w, w + wn);          // `tmp` is reused below
            const std::string& okey = wkey;
            if (o.kind == SBC_OP_CONV_PAIR || o.kind == SBC_OP_CONV_POOL || o.kind == SBC_OP_RES_BLOCK || o.kind == SBC_OP_CHAIN) {
                // the fused kernels read the direct fp16 forms only
                if (f16x2) sbc_pack_conv_weight_f16x2(wkeep.data(), cout, cin, k, (uint16_t*)reserve(okey + "#split", sbc_f16x2_elems(k * k, cin, cout) / 2));
                else sbc_pack_conv_weight_f16(wkeep.data(), cout, cin, k, (uint16_t*)reserve(okey + "#split", (wn + 1) / 2));
            } else if (o.kind != SBC_OP_CONV) {
                memcpy(reserve(o.weight, wn), wkeep.data(), sizeof(float) * wn);
            } else if (d->conv_mode == 1) {
                sbc_pack_conv_weight(wkeep.data(), cout, cin, k, reserve(o.weight, wn));
                if (k == 3) sbc_pack_conv_weight_winograd(wkeep.data(), cout, cin, reserve(o.weight + "#winograd", (size_t)cout * cin * 16));
            } else if (d->conv_mode == 0) {
                sbc_pack_conv_weight_split(wkeep.data(), cout, cin, k, (uint16_t*)reserve(o.weight + "#split", wn * 3 / 2));
                if (k == 3) sbc_pack_conv_weight_winograd_split(wkeep.data(), cout, cin,
                                                               (uint16_t*)reserve(o.weight + "#winograd_split", (size_t)cout * cin * 16 * 3 / 2));
            } else if (f16x2) {
                sbc_pack_conv_weight_f16x2(wkeep.data(), cout, cin, k, (uint16_t*)reserve(o.weight + "#split", sbc_f16x2_elems(k * k, cin, cout) / 2));
                if (k == 3) sbc_pack_conv_weight_winograd_f16x2(wkeep.data(), cout, cin,
                                                               (uint16_t*)reserve(o.weight + "#winograd_split", sbc_f16x2_elems(16, cin, cout) / 2));
            } else {
                sbc_pack_conv_weight_f16(wkeep.data(), cout, cin, k, (uint16_t*)reserve(o.weight + "#split", (wn + 1) / 2));
                if (k == 3) sbc_pack_conv_weight_winograd_f16(wkeep.data(), cout, cin,
                                                             (uint16_t*)reserve(o.weight + "#winograd_split", (size_t)cout * cin * 16 / 2));
            }
        }
        }
        std::vector<std::string> bkeys{o.bias, o.bias2};
        for (const auto& bl : o.blocks) { bkeys.push_back(bl.bias1); bkeys.push_back(bl.bias2); bkeys.push_back(bl.bias3); }
        for (const std::string& bkey : bkeys) {
            if (bkey.empty() || off.count(bkey)) continue;
            const float* bv = find(bkey, cout);
            if (!bv) return fail();
            bv = rounded(bv, cout);
            memcpy(reserve(bkey, cout), bv, sizeof(float) * cout);
        }
    }
    // ---- device memory
    auto hip_fail = [&](hipError_t e, const char* what) {
        set_error("sbc_score_create: %s failed: %s", what, hipGetErrorString(e));
        sbc_score_destroy(s);
        return SBC_ERR_HIP;
    };
    hipError_t e;
    if ((e = hipMalloc((void**)&s->wdev, host.size() * sizeof(float))) != hipSuccess) return hip_fail(e, "hipMalloc(weights)");
    if ((e = hipMemcpy(s->wdev, host.data(), host.size() * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "hipMemcpy(weights)");
    if ((e = hipMalloc((void**)&s->sigmas, d->num_classes * sizeof(float))) != hipSuccess) return hip_fail(e, "hipMalloc(sigmas)");
    if ((e = hipMemcpy(s->sigmas, d->sigmas, d->num_classes * sizeof(float), hipMemcpyHostToDevice)) != hipSuccess) return hip_fail(e, "hipMemcpy(sigmas)");
    if ((e = hipMalloc((void**)&s->labels, B * sizeof(int64_t))) != hipSuccess) return hip_fail(e, "hipMalloc(labels)");
    if ((e = hipMemset(s->labels, 0, B * sizeof(int64_t))) != hipSuccess) return hip_fail(e, "hipMemset(labels)");
    s->slots.assign(s->slot_elems.size(), nullptr);
    for (size_t i = 0; i < s->slot_elems.size(); ++i)
        if ((e = hipMalloc((void**)&s->slots[i], (size_t)B * s->slot_elems[i] * sizeof(float))) != hipSuccess)
            return hip_fail(e, "hipMalloc(activation slot)");
    s->endc.sigmas = s->sigmas; s->endc.labels = s->labels; s->endc.sigma_of_step = nullptr; s->endc.step = nullptr;
    // ---- records (scorenet.ScoreNet.bind)
    size_t n_chains = 0;
    for (const POp& o : s->pops) n_chains += o.kind == SBC_OP_CHAIN;
    s->chains.reserve(n_chains);
    for (const POp& o : s->pops) {
        sbc_op r;
        memset(&r, 0, sizeof(r));
        const Tn& src = s->tensors[o.geom >= 0 ? o.geom : o.src];     // (statistics from tile moments: the image's dims)
        const Tn& dst = s->tensors[o.dst];
        r.kind = o.kind; r.flags = o.flags; r.B = B; r.H = src.h; r.W = src.w;
        r.cin = src.c; r.cout = dst.c; r.ksize = o.ksize; r.dil = o.dil; r.tag = o.tag;
        r.lane = o.lane; r.signal = o.signal; r.wait[0] = o.wait0; r.wait[1] = o.wait1;
        r.in = s->slots[s->tensors[o.src].slot]; r.out = s->slots[dst.slot];
        if (o.moments >= 0) r.aux = s->slots[s->tensors[o.moments].slot];
        auto wp = [&](const std::string& key) -> const void* { return off.count(key) ? s->wdev + off[key] : nullptr; };
        if (o.kind == SBC_OP_CONV_PAIR) {
            r.weight_split = wp(o.weight + "#split");
            r.weight2_split = wp(o.weight2 + "#split");
            if (f16x2) { r.weight_wino_split = wp(o.weight + "#winograd_split"); r.weight2_wino_split = wp(o.weight2 + "#winograd_split"); }   // calibration only
            r.flags |= f16w ? SBC_CONV_F16W : SBC_CONV_F16X2;
        } else if (o.kind == SBC_OP_CONV_POOL) {
            r.weight_split = wp(o.weight + "#split");
            if (f16x2) r.weight_wino_split = wp(o.weight + "#winograd_split");
            r.flags |= f16w ? SBC_CONV_F16W : SBC_CONV_F16X2;
        } else if (o.kind == SBC_OP_CONV_DOWN) {
            r.weight_split = wp(o.weight + "#pool");
            r.weight2_split = wp(o.weight2 + "#pool");
            r.bias2 = wp(o.bias2);
            r.flags |= SBC_CONV_F16X2;
        } else if (o.kind == SBC_OP_CHAIN) {
            sbc_chain ch;
            memset(&ch, 0, sizeof(ch));
            ch.n_blocks = (int32_t)o.blocks.size();
            for (size_t k = 0; k < o.blocks.size(); ++k) {
                ch.type[k] = o.blocks[k].type;
                const POp::Block& bl = o.blocks[k];
                ch.w1[k] = wp(bl.w1 + "#split"); ch.w2[k] = wp(bl.w2 + "#split");
                ch.w1_wino[k] = wp(bl.w1 + "#winograd_split"); ch.w2_wino[k] = wp(bl.w2 + "#winograd_split");
                if (bl.type == SBC_CHAIN_RES) {
                    ch.dil[k] = bl.dil;
                    ch.bias1[k] = (const float*)wp(bl.bias1); ch.bias2[k] = (const float*)wp(bl.bias2);
                    ch.norm1[k] = (const float*)wp(bl.norm1); ch.norm2[k] = (const float*)wp(bl.norm2);
                    if (!bl.w3.empty()) { ch.w3[k] = wp(bl.w3 + "#split"); ch.bias3[k] = (const float*)wp(bl.bias3); }
                }
            }
            s->chains.push_back(ch);
            r.ext = &s->chains.back();
            r.flags |= SBC_CONV_F16X2;
        } else if (o.kind == SBC_OP_RES_BLOCK) {
            r.weight_split = wp(o.weight + "#split");
            r.weight2_split = wp(o.weight2 + "#split");
            r.weight_wino_split = wp(o.weight + "#winograd_split"); r.weight2_wino_split = wp(o.weight2 + "#winograd_split");
            r.bias2 = wp(o.bias2);
            r.norm2 = wp(o.norm2);
            r.flags |= SBC_CONV_F16X2;
        } else if (o.kind != SBC_OP_CONV) {
            r.weight = wp(o.weight);
        } else if (d->conv_mode == 1) {
            r.weight = wp(o.weight);
            if (o.ksize == 3 && o.dil == 1) r.weight_wino = wp(o.weight + "#winograd");
        } else {
            r.weight_split = wp(o.weight + "#split");
            if (o.ksize == 3 && o.dil == 1) r.weight_wino_split = wp(o.weight + "#winograd_split");
            if (f16w) r.flags |= SBC_CONV_F16W;
            if (f16x2) r.flags |= SBC_CONV_F16X2;
        }
        if ((d->conv_mode == 0 || d->conv_mode == 1) && o.kind == SBC_OP_CONV && (o.flags & SBC_PRO_ELU)) r.flags |= SBC_PRO_ELU_ACC;   // scorenet.bind
        if (!o.bias.empty()) r.bias = wp(o.bias);
        if (o.stats >= 0) r.stats = s->slots[s->tensors[o.stats].slot];
        if (!o.norm_key.empty()) r.stats = wp(o.norm_key);
        if (o.res1 >= 0) r.res1 = s->slots[s->tensors[o.res1].slot];
        if (o.res2 >= 0) r.res2 = s->slots[s->tensors[o.res2].slot];
        if (o.up >= 0) { r.up = s->slots[s->tensors[o.up].slot]; r.up_h = s->tensors[o.up].h; r.up_w = s->tensors[o.up].w; }
        if (o.kind == SBC_OP_END_CONV) r.ext = &s->endc;
        s->ops.push_back(r);
    }
    const int rc = sbc_plan_create(s->ops.data(), (int32_t)s->ops.size(), &s->plan);
    if (rc) { sbc_score_destroy(s); return rc; }
    if (f16x2) {
        // per-layer activation scales (include/sbc_hip.h: sbc_f16x2_calibrate; scorenet.ScoreNet._ensure_calibrated): one pass over
        // the fixed calibration input in the first sample of the x buffer
        const size_t n = (size_t)nt * nr * d->channels;
        std::vector<float> pat(n);
        int rc2 = sbc_f16x2_calibration_input(pat.data(), (int64_t)n);
        if (!rc2 && hipMemcpy(s->slots[s->tensors[s->x_t].slot], pat.data(), n * sizeof(float), hipMemcpyHostToDevice) != hipSuccess) {
            set_error("sbc_score_create: upload of the calibration input failed");
            rc2 = SBC_ERR_HIP;
        }
        if (!rc2) rc2 = sbc_f16x2_calibrate(s->ops.data(), (int32_t)s->ops.size(), nullptr);
        if (rc2) { sbc_score_destroy(s); return rc2; }
    }
    *out = s;
    return SBC_OK;
}

int sbc_score_buffers(sbc_score* s, float** x, float** out, int64_t** labels) {
    SBC_REQUIRE(s, "sbc_score_buffers: handle is NULL");
    if (x) *x = s->slots[s->tensors[s->x_t].slot];
    if (out) *out = s->slots[s->tensors[s->out_t].slot];
    if (labels) *labels = s->labels;
    return SBC_OK;
}

int sbc_score_ops(sbc_score* s, const sbc_op** ops, int32_t* n_ops) {
    SBC_REQUIRE(s && ops && n_ops, "sbc_score_ops: bad arguments");
    *ops = s->ops.data();
    *n_ops = (int32_t)s->ops.size();
    return SBC_OK;
}

int sbc_score_level_source(sbc_score* s, const float* sigma_of_step, const int32_t* step) {
    SBC_REQUIRE(s, "sbc_score_level_source: handle is NULL");
    SBC_REQUIRE((sigma_of_step == nullptr) == (step == nullptr), "sbc_score_level_source: both pointers or neither");
    s->endc.sigma_of_step = sigma_of_step;
    s->endc.step = step;
    s->endc.labels = step ? nullptr : s->labels;
    // the score plan holds a copy of the end-conv extension: rebuild it
    sbc_plan_destroy(s->plan);
    s->plan = nullptr;
    return sbc_plan_create(s->ops.data(), (int32_t)s->ops.size(), &s->plan);
}

int sbc_score_forward(sbc_score* s, void* stream) {
    SBC_REQUIRE(s && s->plan, "sbc_score_forward: handle is NULL");
    return sbc_plan_run(s->plan, stream, 1, 0);
}

void sbc_score_destroy(sbc_score* s) {
    if (!s) return;
    if (s->plan) sbc_plan_destroy(s->plan);
    for (float* p : s->slots) if (p) (void)hipFree(p);
    if (s->wdev) (void)hipFree(s->wdev);
    if (s->sigmas) (void)hipFree(s->sigmas);
    if (s->labels) (void)hipFree(s->labels);
    delete s;
}

}  // extern "C"
