// network.h -- static graph of the build-defined RetinaFace network,
// its weights, workspace plan and executor.  Replaces the Triton model repository + remote
// inference of the reference (face_detection.rs:234-312).
#pragma once
#include <string>
#include <vector>

#include "kernels.h"

namespace rfd {

enum LayerKind { LK_CONV = 0, LK_DEPTHWISE = 1, LK_FIRST3X3 = 2, LK_CONV0 = 3 };

struct Layer {
    std::string name;
    int cin, cout, kh, kw, stride, pad; // logical shape (depthwise: cin = 1, cout = channels)
    int kind;         // LayerKind
    int cin_d, cout_d; // device shape: channels zero-padded to the granularity of the MFMA conv kernel
    int has_affine; // per-channel scale2/shift2 (the BN+ReLU that follows a residual add / the pool)
    float gain;     // synthetic-init gain on the He std
    size_t w_off;   // element offset into the bf16 weight buffer (first column of this layer's block)
    size_t ldw;     // device row pitch in elements (> kh*kw*cin when a shortcut conv shares the rows)
    size_t w_elems; // device elements (conv0 is stored K-padded: 64*7*32)
    size_t b_off;   // float offset of the bias [cout] (consecutive layers are contiguous: N-fused convs)
    size_t a_off;   // float offset of the post-add affine: scale [cout] then shift [cout] (has_affine only)
};

enum OpKind { OP_CONV0 = 0, OP_POOL = 1, OP_CONV = 2, OP_STEM = 3, OP_DW = 4, OP_FIRST = 5, OP_B2B = 6 };

struct Op {
    int kind;
    int layer;
    int in, out, out2, outf, res; // tensor ids, -1 = none
    int in2, layer2;              // fused second K segment (1x1 shortcut conv on tensor in2), -1 = none
    int in_affine;                // layer whose post-add affine (+ReLU) is applied to this op's INPUT, -1 = none
    int layer_n2;                 // second conv fused along N (same input and geometry; its rows follow), -1 = none
    int x_coff;                   // the input is the channel slice [x_coff, x_coff+cin) of tensor `in`
    int y_split, y_split_add;     // output channels >= y_split are stored y_split_add channels further
    int n_valid;                  // only output channels < n_valid are stored (padded weight rows)
    int branch;                   // 0 = main stream; 1, 2 = independent side chains (SSH + head of stride 32 / 16)
    int layer_b, out_b;           // OP_B2B: the NEXT unit's conv1 (applied to relu(affine(out))) and its output tensor
    int relu, res_up2, res_post, head_softmax, y_coff;
};

struct TensorDesc {
    int C, H, W;    // C = device channels (>= C_logical, zero padded)
    int C_logical;
    int is_f32;
    int buffer;     // workspace buffer id (assigned by plan())
    int first, last; // op index range in which the tensor is live
    size_t bytes_per_image() const { return (size_t)C * H * W * (is_f32 ? 4 : 2); }
};

struct Graph {
    int backbone = 0, net_h = 0, net_w = 0;
    std::vector<Layer> layers;
    std::vector<Op> ops;
    std::vector<TensorDesc> tensors;
    int input = -1;        // NHWC4 bf16 network input
    int heads[3] = {-1, -1, -1}; // f32 [h][w][32] per level (stride 32,16,8)
    size_t w_total = 0, b_total = 0, a_total = 0;
    std::vector<size_t> buffer_bytes_per_image; // workspace plan

    int build(int backbone, int net_w, int net_h);
    double macs_per_image() const; // conv multiply-accumulates (Appendix B: 44.265e9 at 640x640 R50)
    double layer_macs(int op_index) const;

  private:
    int add_tensor(int C, int H, int W, int f32 = 0);
    int add_layer(const std::string &name, int cin, int cout, int k, int stride, int pad, float gain,
                  int has_affine, int extra_k = 0, int parent = -1, int col_off = 0, int kind = LK_CONV,
                  int cout_dev = 0);
    int add_conv(int layer, int in, int out, int relu, int res = -1, int out2 = -1, int outf = -1);
    void build_r50();
    void build_mnet025();
    void plan();
};

struct Network {
    Graph g;
    int max_batch = 0;
    bool weights_ready = false;
    bf16_t *d_w = nullptr;
    float *d_b = nullptr;
    bf16_t *d_zero = nullptr;
    int *d_fail = nullptr;    // device fault word of the owning context (bounded spins of the ring convolutions), or null
    std::vector<void *> d_buffers;
    bool profiling = false;
    int force_tile = 0; // test hook: 0 heuristic, 1 = 128-row tiles only, 2 = 256x128 wherever Cout % 128 == 0
    std::vector<hipEvent_t> ev; // 2 per op when profiling
    std::vector<float> op_ms;   // last profiled run
    int prof_first = 0, prof_last = -1; // op range of the last run

    // f32 parity mode (rfd_config.precision = RFD_PRECISION_F32; kernels_f32.hip): f32 copies of the weights (same element
    // offsets as d_w) and of the workspace buffers (same plan, 4 bytes per element); run() then walks the op list with the f32
    // kernels on the caller's stream -- no batch split, no hipGraph, no persistent kernels.  Heads and the network input keep
    // their ordinary buffers (f32 / exact bf16), so preprocess and decode are shared with the bf16 path.
    int precision = 0;
    float *d_w32 = nullptr;
    void *d_stem32 = nullptr; // conv0 activation of the (unfused) f32 stem
    std::vector<void *> d_buffers32;
    void *tensor_ptr32(int t, int batch_off = 0) const
    {
        if (g.tensors[t].is_f32 || t == g.input) return tensor_ptr(t, batch_off);
        const int bi = g.tensors[t].buffer;
        return (char *)d_buffers32[bi] + (size_t)batch_off * g.buffer_bytes_per_image[bi] * 2;
    }
    int run_f32(int B, hipStream_t s, int first_op, int last_op, int batch_off);
    int create(int backbone, int net_w, int net_h, int max_batch, int precision = 0);
    void destroy();
    int init_synthetic(uint64_t seed, hipStream_t s);
    int get_layer(int idx, float *w, float *bias, hipStream_t s);
    int set_layer(int idx, const float *w, const float *bias, hipStream_t s);
    int get_affine(int idx, float *scale, float *shift, hipStream_t s);
    int set_affine(int idx, const float *scale, const float *shift, hipStream_t s);
    // Images [batch_off, ...) of a part of a split pass live at a fixed PER-BUFFER offset: tensors that share a buffer have
    // different per-image sizes, so per-tensor image offsets of the parts would overlap.  The input and head
    // tensors own exactly-sized buffers (plan()), so for them this is the ordinary image-major layout.
    void *tensor_ptr(int t, int batch_off = 0) const
    {
        const int bi = g.tensors[t].buffer;
        void *base = (head_parity && bi < (int)d_alt.size() && d_alt[bi]) ? d_alt[bi] : d_buffers[bi];
        return (char *)base + (size_t)batch_off * g.buffer_bytes_per_image[bi];
    }
    // second copy of the three head buffers: with calls overlapped across steps (rfd_detect_batch_device, async = 2)
    // the chains of call i+1 write heads[parity ^ 1] while decode of call i still reads heads[parity]
    std::vector<void *> d_alt;
    int head_parity = 0;
    int ensure_alt_heads();
    int run(int B, hipStream_t s, int first_op = 0, int last_op = -1, int batch_off = 0, int part = 0);
    int run_split(int B, hipStream_t s); // whole pass; splits the batch over several streams when it pays
    int split_body(int B, int P, hipStream_t s);
    int tune_streams(int B, int P, hipStream_t s);
    void assign_streams(int a, int b, int c);
    static constexpr int kPool = 8;
    hipStream_t pool[kPool] = {};
    bool tuned = false, tune = true;
    float tuned_ms = 0.f;
    int tuned_a = -1, tuned_b = -1;
    // whole op list replayed from a hipGraph captured per batch size (removes ~5 us of launch gap per kernel;
    // matters at batch 1, where the network is launch-bound); falls back to run() while profiling
    int run_graphed(int B, hipStream_t s);
    std::vector<hipGraphExec_t> graph_exec; // indexed by B
    std::vector<char> warmed;               // eager run done for this B (one-time kernel attribute set-up)
    bool use_graph = true;
    // side streams: the stride-32 / stride-16 SSH + head chains have tiny grids and no dependency on the stride-8
    // chain, so they run concurrently with it (fork after their FPN input, join at the end of the pass)
    static constexpr int kMaxParts = 4;
    hipStream_t side[kMaxParts][2] = {}; // [part][branch - 1]
    hipEvent_t ev_fork[kMaxParts][2] = {}, ev_join[kMaxParts][2] = {};
    bool multi_stream = true;
    // batch split: contiguous parts of a batch are independent chains over disjoint slices of the same image-major
    // workspace; running them on their own streams lets the tail of one part's kernels overlap the others' (grid
    // quantisation at B = 32 costs ~10 %: B = 256 measures 6.0 k img/s vs 5.5 k unsplit).  A split pass is always
    // launched eagerly: replayed from a hipGraph it measured SLOWER than the unsplit graph (5.07 k vs 5.46 k img/s at
    // B = 32; eager split 6.07 k), and launch gaps are hidden at these batch sizes anyway.
    hipStream_t part_stream[kMaxParts] = {}; // [0]: only the cross-call overlap mode runs part 0 off the caller's stream
    hipEvent_t ev_part_fork = nullptr, ev_part_join[kMaxParts] = {};
    // chain phase shift: part 1 starts only after part 0 has finished op `chain_shift_op` of the same pass, so that the two
    // chains do not run the SAME kernel class side by side (compute-bound next to compute-bound, memory-bound next to
    // memory-bound) but stay a few ops apart; -1 = both start together
    int chain_shift_op = -1;
    hipEvent_t ev_shift = nullptr;
    int co_running = 0;      // set while the parts of a split pass are being enqueued
    int split_min_part = 4;  // fewest images a part may hold (B = 8: 4.02 k img/s split vs 3.86 k as one graph; B <= 6: graph wins or ties)
    int split_max_parts = 2; // parts = clamp(B / split_min_part, 1, split_max_parts)
    int num_parts(int B) const
    {
        if (profiling || split_min_part < 1 || precision != 0) return 1;
        return std::max(1, std::min(std::min(split_max_parts, kMaxParts), B / split_min_part));
    }
    int collect_profile(); // after the stream has drained
};

} // namespace rfd
