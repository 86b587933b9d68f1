"""torch-CPU f32 reference executor for the build-defined network (test infrastructure).

Walks the op list the library exposes (rfd_graph_*) and evaluates every op with torch functional
ops on the SAME weights read back from the device, rounding activations to bf16 exactly where the
HIP kernels do (every stored tensor except the f32 heads).  It is an independent implementation of
the convolutions (torch/oneDNN), not the reference's network: the reference has none (the model
lives on a Triton server that is not in the repo) -- CNN parity is "unpinned" (DESIGN.md)."""
import numpy as np
import torch
import torch.nn.functional as F


_ROUND = [True]  # TorchRef(round_bf16=False) evaluates the same op list WITHOUT the bf16 roundings: the f32 parity mode's reference
_ACC64 = [False]  # TorchRef(acc64=True): convolutions summed in f64 and rounded to f32 ONCE per output, elementwise steps in the
                  # device's order (csrc/kernels_f32.hip since round 4) -- two such evaluations agree to the ulp


def bf16r(t):
    return t.to(torch.bfloat16).to(torch.float32) if _ROUND[0] else t


def conv(x, w, b, stride=1, padding=0, groups=1, x2=None, w2=None, b2=None, stride2=1):
    """conv2d + bias (+ a second 1x1 conv fused as K segment).  acc64: both sums in ONE f64 accumulator, rounded to f32 once,
    then + (b + b2) in f32 -- the device kernel's order (conv_f32_kernel: `(float)acc + bias`, bias = bias + bias2)."""
    if not _ACC64[0]:
        v = F.conv2d(x, w, b, stride=stride, padding=padding, groups=groups)
        if x2 is not None:
            v = v + F.conv2d(x2, w2, b2, stride=stride2)
        return v
    s = F.conv2d(x.double(), w.double(), None, stride=stride, padding=padding, groups=groups)
    bias = b
    if x2 is not None:
        s = s + F.conv2d(x2.double(), w2.double(), None, stride=stride2)
        bias = b + b2
    return s.float() + bias.view(1, -1, 1, 1)


def affine_in(x, s, t):
    """the producer unit's BN + ReLU on a conv operand.  The device applies it as ONE fused multiply-add (fmaf) in the f32 parity
    mode; x * s is exact in f64, so the f64 expression rounded once reproduces fmaf (but for double-rounding cases ~2^-29)."""
    if _ACC64[0]:
        return F.relu((x.double() * s.double().view(1, -1, 1, 1) + t.double().view(1, -1, 1, 1)).float())
    return F.relu(x * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1))


class TorchRef:
    def __init__(self, graph, det, round_bf16=True, acc64=False, round_ops=None):
        """graph: rfd_hip.Graph; det: rfd_hip.RetinaFaceDetection with initialised weights.
        round_ops: set of op indices whose stored tensors are rounded to bf16 (all others stay f32) -- the probe of the bf16 error
        budget (tools/error_budget.py); None: every op rounds, or none, as round_bf16 says."""
        self.g = graph
        self.round_bf16 = round_bf16
        self.acc64 = acc64
        self.round_ops = round_ops
        assert not (acc64 and (round_bf16 or round_ops)), "acc64 is the reference of the f32 parity mode"
        self.w, self.b, self.aff = [], [], []
        for i, L in enumerate(graph.layers):
            w, b = det.get_layer(i, L)
            self.w.append(torch.from_numpy(w).permute(0, 3, 1, 2).contiguous())  # [cout][cin][kh][kw]
            self.b.append(torch.from_numpy(b))
            if L.has_affine:
                s, t = det.get_affine(i, L.cout)
                self.aff.append((torch.from_numpy(s), torch.from_numpy(t)))
            else:
                self.aff.append(None)

    def run_op(self, i, tensors):
        """tensors: dict id -> NCHW f32 torch tensor (bf16-rounded values).  Evaluates op i."""
        g = self.g
        o = g.ops[i]
        L = g.layers[o.layer]
        x = tensors[o.in_]
        if o.kind == 2 and x.shape[1] != L.cin:  # the input is a channel slice of a wider tensor
            x = x[:, o.x_coff:o.x_coff + L.cin]
        if o.kind == 0:  # conv0 7x7/2 on R,G,B (+ zero 4th channel) + bias + relu
            v = conv(x[:, :3], self.w[o.layer], self.b[o.layer], stride=2, padding=3)
            tensors[o.out] = bf16r(F.relu(v))
            return
        if o.kind == 5:  # first 3x3/2 conv on R,G,B + bias + relu (MobileNet)
            tensors[o.out] = bf16r(F.relu(F.conv2d(x[:, :3], self.w[o.layer], self.b[o.layer], stride=2, padding=1)))
            return
        if o.kind == 4:  # depthwise 3x3 + bias + relu
            tensors[o.out] = bf16r(F.relu(F.conv2d(x, self.w[o.layer], self.b[o.layer], stride=L.stride, padding=1,
                                                   groups=L.cout)))
            return
        if o.kind == 3:  # fused stem: conv0 + bias + relu (bf16) -> maxpool 3x3/2 pad 1 -> affine + relu
            v = bf16r(F.relu(conv(x[:, :3], self.w[o.layer], self.b[o.layer], stride=2, padding=3)))
            s, t = self.aff[o.layer]
            tensors[o.out] = bf16r(F.relu(F.max_pool2d(v, 3, 2, 1) * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)))
            return
        if o.kind == 1:  # maxpool 3x3/2 pad 1, then affine + relu
            v = F.max_pool2d(x, 3, 2, 1)
            s, t = self.aff[o.layer]
            tensors[o.out] = bf16r(F.relu(v * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)))
            return
        if o.in_affine >= 0:  # BN+ReLU of the producer unit, applied to this conv's input (rounded to bf16)
            s, t = self.aff[o.in_affine]
            x = bf16r(affine_in(x, s, t))
        wgt, bias = self.w[o.layer], self.b[o.layer]
        if o.layer_n2 >= 0:  # sibling conv on the same input fused along N: its output channels follow
            wgt = torch.cat([wgt, self.w[o.layer_n2]], 0)
            bias = torch.cat([bias, self.b[o.layer_n2]], 0)
        if o.layer2 >= 0:  # the 1x1 shortcut conv fused as a second K segment of the same GEMM
            L2 = g.layers[o.layer2]
            v = conv(x, wgt, bias, stride=L.stride, padding=L.pad, x2=tensors[o.in2], w2=self.w[o.layer2], b2=self.b[o.layer2], stride2=L2.stride)
        else:
            v = conv(x, wgt, bias, stride=L.stride, padding=L.pad)
        r = None
        if o.res >= 0:
            r = tensors[o.res]
            if o.res_up2:
                r = F.interpolate(r, scale_factor=2, mode="nearest")
            if not o.res_post:
                v = v + r
        if o.out >= 0:
            y = F.relu(v) if o.relu else v
            if r is not None and o.res_post:
                y = y + r
            y = bf16r(y)
            td = g.tensors[o.out]
            nout = y.shape[1]
            if td.channels_logical != nout:  # SSH concat buffer: channel n -> y_coff + n (+ y_split_add if n >= y_split)
                if o.out not in tensors:
                    tensors[o.out] = torch.zeros(x.shape[0], td.channels_logical, td.height, td.width)
                s = min(o.y_split, nout)
                tensors[o.out][:, o.y_coff:o.y_coff + s] = y[:, :s]
                if s < nout:
                    tensors[o.out][:, o.y_coff + o.y_split_add + s:o.y_coff + o.y_split_add + nout] = y[:, s:]
            else:
                tensors[o.out] = y
        if o.out2 >= 0:
            s, t = self.aff[o.layer]
            tensors[o.out2] = bf16r(F.relu(v * s.view(1, -1, 1, 1) + t.view(1, -1, 1, 1)))
        if o.kind == 6:  # back to back: the next unit's conv1 on relu(affine(bf16 raw)), bias + relu
            s, tt = self.aff[o.layer]
            if o.out >= 0:
                a = bf16r(affine_in(tensors[o.out], s, tt))
            else:            # the last unit of a stage: only the activated output exists; conv1 reads it as stored
                a = tensors[o.out2]
            tensors[o.out_b] = bf16r(F.relu(conv(a, self.w[o.layer_b], self.b[o.layer_b])))
        if o.outf >= 0:
            if o.head_softmax:  # channels 0,1 = bg(a), 2,3 = fg(a): softmax over the pairs (a, A+a)
                pr = torch.softmax(torch.stack([v[:, 0:2], v[:, 2:4]], 0), 0)
                v = torch.cat([pr[0], pr[1], v[:, 4:]], 1)
            tensors[o.outf] = v

    def forward(self, x_nchw4, upto=None):
        """x: [n,4,H,W] f32 (R,G,B,0).  Returns dict of all tensors (NCHW f32)."""
        tensors = {next(i for i, t in enumerate(self.g.tensors) if t.is_input): x_nchw4}
        last = len(self.g.ops) - 1 if upto is None else upto
        _ROUND[0] = self.round_bf16
        _ACC64[0] = self.acc64
        try:
            with torch.no_grad():
                for i in range(last + 1):
                    if self.round_ops is not None:
                        _ROUND[0] = i in self.round_ops
                    self.run_op(i, tensors)
        finally:
            _ROUND[0] = True
            _ACC64[0] = False
        return tensors

    def heads(self, tensors):
        """Head tensors in the reference's 9-tensor NCHW contract."""
        out = []
        for lvl in (1, 2, 3):
            tid = next(i for i, t in enumerate(self.g.tensors) if t.head_level == lvl)
            v = tensors[tid]
            out += [v[:, 0:4].numpy().copy(), v[:, 4:12].numpy().copy(), v[:, 12:32].numpy().copy()]
        return out


def nchw_to_dev(t, is_f32=False, channels=None):
    """NCHW f32 torch tensor -> device layout (NHWC; bf16 bits as uint16, or f32), channels zero-padded."""
    if channels is not None and channels > t.shape[1]:
        t = torch.cat([t, torch.zeros(t.shape[0], channels - t.shape[1], t.shape[2], t.shape[3])], 1)
    a = t.permute(0, 2, 3, 1).contiguous()
    if is_f32:
        return a.numpy().astype(np.float32)
    return a.to(torch.bfloat16).view(torch.int16).numpy().view(np.uint16)


def dev_to_nchw(a, is_f32=False, channels=None):
    """Inverse of nchw_to_dev; `channels` = logical channel count (the zero padding is checked and dropped)."""
    t = _dev_to_nchw(a, is_f32)
    if channels is not None and channels < t.shape[1]:
        assert float(t[:, channels:].abs().max()) == 0.0, "channel padding is not zero"
        t = t[:, :channels].contiguous()
    return t


def _dev_to_nchw(a, is_f32=False):
    if is_f32:
        return torch.from_numpy(a.astype(np.float32)).permute(0, 3, 1, 2).contiguous()
    t = torch.from_numpy(a.view(np.int16).copy()).view(torch.bfloat16).to(torch.float32)
    return t.permute(0, 3, 1, 2).contiguous()
