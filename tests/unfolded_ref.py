"""An f32 torch model of the RetinaFace-R50 of SURVEY.md Appendix B written DIRECTLY from that description, with
explicit inference-mode BatchNorm layers -- i.e. in the unfolded form in which checkpoints are published.  It shares
no code with the device graph builder or with tests/torch_ref.py (which walks the library's own op list): agreement
between this model and the device network loaded through rfd_hip.convert.import_unfolded() checks topology, tap
points and the BN folding convention together (test infrastructure)."""
import numpy as np
import torch
import torch.nn.functional as F

UNITS = {1: 3, 2: 4, 3: 6, 4: 3}
MID = {1: 64, 2: 128, 3: 256, 4: 512}
EPS = 2e-5


def make_params(seed):
    """Random parameters with sane statistics (He-scaled convs, BN statistics near identity) under our key names."""
    rng = np.random.default_rng(seed)
    P = {}

    def conv(name, cout, cin, k, gain=1.0):
        P[name + "_weight"] = (rng.normal(0, gain * np.sqrt(2.0 / (cin * k * k)), size=(cout, cin, k, k))).astype(np.float32)

    def bn(name, c):
        P[name + "_gamma"] = rng.uniform(0.6, 1.4, c).astype(np.float32)
        P[name + "_beta"] = rng.normal(0, 0.15, c).astype(np.float32)
        P[name + "_mean"] = rng.normal(0, 0.2, c).astype(np.float32)
        P[name + "_var"] = rng.uniform(0.5, 1.5, c).astype(np.float32)

    conv("conv0", 64, 3, 7, gain=1.0 / 90.0)  # raw 0..255 input
    bn("conv0_bn", 64)
    cin = 64
    for s in (1, 2, 3, 4):
        m = MID[s]
        for u in range(1, UNITS[s] + 1):
            p = "stage%d_unit%d" % (s, u)
            bn(p + "_bn1", cin)
            conv(p + "_conv1", m, cin, 1); bn(p + "_conv1_bn", m)
            conv(p + "_conv2", m, m, 3); bn(p + "_conv2_bn", m)
            conv(p + "_conv3", 4 * m, m, 1, gain=0.5)
            if u == 1:
                conv(p + "_sc", 4 * m, cin, 1, gain=0.5)
            cin = 4 * m
    bn("bn1", 2048)
    for nm, ci in (("fpn_lat3", 2048), ("fpn_lat2", 1024), ("fpn_lat1", 512)):
        conv(nm, 256, ci, 1); bn(nm + "_bn", 256)
    for nm in ("fpn_aggr2", "fpn_aggr1"):
        conv(nm, 256, 256, 3); bn(nm + "_bn", 256)
    for st in (32, 16, 8):
        for nm, co, ci in (("conv1", 128, 256), ("ctx1", 64, 256), ("ctx2", 64, 64), ("ctx3a", 64, 64), ("ctx3b", 64, 64)):
            conv("ssh%d_%s" % (st, nm), co, ci, 3); bn("ssh%d_%s_bn" % (st, nm), co)
        for nm, co in (("cls", 4), ("bbox", 8), ("lmk", 20)):
            conv("head%d_%s" % (st, nm), co, 256, 1, gain=0.5)
            P["head%d_%s_bias" % (st, nm)] = rng.normal(0, 0.1, co).astype(np.float32)
    return P


def forward(P, x):
    """x: [n,3,H,W] f32, R,G,B raw 0..255 -> the 9 head tensors of the Triton contract (cls soft-maxed)."""
    T = {k: torch.from_numpy(v) for k, v in P.items()}

    def bn(y, k):
        sh = (1, -1, 1, 1)
        return (y - T[k + "_mean"].view(sh)) / torch.sqrt(T[k + "_var"].view(sh) + EPS) * T[k + "_gamma"].view(sh) + T[k + "_beta"].view(sh)

    def cbr(y, k, stride=1, relu=True):
        w = T[k + "_weight"]
        y = bn(F.conv2d(y, w, None, stride, w.shape[2] // 2), k + "_bn")
        return F.relu(y) if relu else y

    y = F.max_pool2d(cbr(x, "conv0", 2), 3, 2, 1)
    taps = {}
    for s in (1, 2, 3, 4):
        for u in range(1, UNITS[s] + 1):
            p = "stage%d_unit%d" % (s, u)
            act = F.relu(bn(y, p + "_bn1"))
            if u == 1:
                taps[s] = act  # the activated input of a stage = the feature map of the previous one
            stride = 2 if (u == 1 and s > 1) else 1
            t = cbr(act, p + "_conv1")
            t = cbr(t, p + "_conv2", stride)
            t = F.conv2d(t, T[p + "_conv3_weight"])
            y = t + (F.conv2d(act, T[p + "_sc_weight"], None, stride) if u == 1 else y)
    c1, c2, c3 = taps[3], taps[4], F.relu(bn(y, "bn1"))  # strides 8, 16, 32
    up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
    p3 = cbr(c3, "fpn_lat3")
    p2 = cbr(cbr(c2, "fpn_lat2") + up(p3), "fpn_aggr2")
    p1 = cbr(cbr(c1, "fpn_lat1") + up(p2), "fpn_aggr1")
    outs = []
    for st, f in ((32, p3), (16, p2), (8, p1)):
        k = "ssh%d_" % st
        c = cbr(f, k + "ctx1")
        o = F.relu(torch.cat([cbr(f, k + "conv1", relu=False), cbr(c, k + "ctx2", relu=False),
                              cbr(cbr(c, k + "ctx3a"), k + "ctx3b", relu=False)], 1))
        cls = F.conv2d(o, T["head%d_cls_weight" % st], T["head%d_cls_bias" % st])
        n, _, h, w = cls.shape
        cls = torch.softmax(cls.view(n, 2, 2, h, w), 1).view(n, 4, h, w)  # pairs (a, A + a), A = 2
        outs += [cls, F.conv2d(o, T["head%d_bbox_weight" % st], T["head%d_bbox_bias" % st]),
                 F.conv2d(o, T["head%d_lmk_weight" % st], T["head%d_lmk_bias" % st])]
    return [t.numpy() for t in outs]


class FoldedF32:
    """The same network as `forward`, as an inference server would run it on a CPU: BatchNorm folded into the preceding
    convolution (conv + bias + ReLU), plain f32, channels_last, torch.inference_mode.  Only the pre-activation BN1 of a
    unit stays an explicit per-channel affine (it follows the residual add).  Used by bench.py's cpu_baseline leg as the
    stand-in for the reference's Triton-CPU backend; `forward(P, x)` above stays the checker of the tests."""

    def __init__(self, P):
        self.w, self.b, self.aff = {}, {}, {}

        def fold(name, with_bn=True):
            w = torch.from_numpy(P[name + "_weight"]).float()
            if with_bn:
                s = torch.from_numpy(P[name + "_bn_gamma"] / np.sqrt(P[name + "_bn_var"] + EPS)).float()
                self.b[name] = torch.from_numpy(P[name + "_bn_beta"]).float() - torch.from_numpy(P[name + "_bn_mean"]).float() * s
                w = w * s.view(-1, 1, 1, 1)
            else:
                self.b[name] = torch.from_numpy(P[name + "_bias"]).float() if (name + "_bias") in P else None
            self.w[name] = w.contiguous(memory_format=torch.channels_last)

        def affine(name):
            s = torch.from_numpy(P[name + "_gamma"] / np.sqrt(P[name + "_var"] + EPS)).float()
            t = torch.from_numpy(P[name + "_beta"]).float() - torch.from_numpy(P[name + "_mean"]).float() * s
            self.aff[name] = (s.view(1, -1, 1, 1), t.view(1, -1, 1, 1))

        fold("conv0")
        for s in (1, 2, 3, 4):
            for u in range(1, UNITS[s] + 1):
                p = "stage%d_unit%d" % (s, u)
                affine(p + "_bn1")
                fold(p + "_conv1"); fold(p + "_conv2"); fold(p + "_conv3", False)
                if u == 1:
                    fold(p + "_sc", False)
        affine("bn1")
        for nm in ("fpn_lat3", "fpn_lat2", "fpn_lat1", "fpn_aggr2", "fpn_aggr1"):
            fold(nm)
        for st in (32, 16, 8):
            for nm in ("conv1", "ctx1", "ctx2", "ctx3a", "ctx3b"):
                fold("ssh%d_%s" % (st, nm))
            for nm in ("cls", "bbox", "lmk"):
                fold("head%d_%s" % (st, nm), False)

    def conv(self, y, k, stride=1, relu=True):
        w = self.w[k]
        y = F.conv2d(y, w, self.b[k], stride, w.shape[2] // 2)
        return F.relu_(y) if relu else y

    def act(self, y, k):
        s, t = self.aff[k]
        return F.relu_(y * s + t)

    @torch.inference_mode()
    def forward(self, x):
        """x: [n,3,H,W] f32 -> the 9 head tensors (torch, cls soft-maxed)."""
        y = F.max_pool2d(self.conv(x.contiguous(memory_format=torch.channels_last), "conv0", 2), 3, 2, 1)
        taps = {}
        for s in (1, 2, 3, 4):
            for u in range(1, UNITS[s] + 1):
                p = "stage%d_unit%d" % (s, u)
                a = self.act(y, p + "_bn1")
                if u == 1:
                    taps[s] = a
                stride = 2 if (u == 1 and s > 1) else 1
                t = self.conv(self.conv(self.conv(a, p + "_conv1"), p + "_conv2", stride), p + "_conv3", relu=False)
                y = t + (self.conv(a, p + "_sc", stride, relu=False) if u == 1 else y)
        c1, c2, c3 = taps[3], taps[4], self.act(y, "bn1")
        up = lambda t: F.interpolate(t, scale_factor=2, mode="nearest")
        p3 = self.conv(c3, "fpn_lat3")
        p2 = self.conv(self.conv(c2, "fpn_lat2") + up(p3), "fpn_aggr2")
        p1 = self.conv(self.conv(c1, "fpn_lat1") + up(p2), "fpn_aggr1")
        outs = []
        for st, f in ((32, p3), (16, p2), (8, p1)):
            k = "ssh%d_" % st
            c = self.conv(f, k + "ctx1")
            o = F.relu_(torch.cat([self.conv(f, k + "conv1", relu=False), self.conv(c, k + "ctx2", relu=False),
                                   self.conv(self.conv(c, k + "ctx3a"), k + "ctx3b", relu=False)], 1))
            cls = self.conv(o, "head%d_cls" % st, relu=False)
            n, _, h, w = cls.shape
            outs += [torch.softmax(cls.reshape(n, 2, 2, h, w), 1).reshape(n, 4, h, w),
                     self.conv(o, "head%d_bbox" % st, relu=False), self.conv(o, "head%d_lmk" % st, relu=False)]
        return outs
