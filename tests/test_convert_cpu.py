"""rfd_hip.convert: the plan that maps unfolded (conv + BatchNorm) parameters onto the folded device layers."""
import numpy as np
import pytest

import unfolded_ref


def test_plan_covers_every_layer_and_parameter(rfd):
    from rfd_hip import convert
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    plan = convert.layer_plan(g)
    assert len(plan) == g.num_layers
    kinds = {e["name"]: e["kind"] for e in plan}
    assert kinds["conv0"] == "bn" and kinds["stage3_unit4_conv2"] == "bn" and kinds["ssh16_ctx3b"] == "bn"
    assert kinds["stage2_unit1_conv3"] == "plain" and kinds["stage2_unit1_sc"] == "plain" and kinds["head8"] == "head"
    aff = {e["name"]: e["affine_bn"] for e in plan if e["affine_bn"]}
    assert aff["conv0"] == "stage1_unit1_bn1" and aff["stage1_unit3_conv3"] == "stage2_unit1_bn1"
    assert aff["stage3_unit2_conv3"] == "stage3_unit3_bn1" and aff["stage4_unit3_conv3"] == "bn1"
    assert len(aff) == 1 + 16                                  # conv0 + one per residual unit

    # the keys the importer reads are exactly the keys of the independent Appendix-B model, and the alias table names
    # every one of them
    P = unfolded_ref.make_params(0)

    class Rec:
        def set_layer(self, i, w, b):
            L = g.layers[i]
            assert w.shape == (L.cout, L.kh, L.kw, L.cin) and b.shape == (L.cout,)

        def set_affine(self, i, s, t):
            assert s.shape == t.shape == (g.layers[i].cout,)

    used = convert.import_unfolded(Rec(), g, P)
    assert used == set(P.keys())
    assert set(convert.INSIGHTFACE_R50_ALIASES.keys()) == set(P.keys())
    assert len(set(convert.INSIGHTFACE_R50_ALIASES.values())) == len(P)
    back = convert.rename({v: P[k] for k, v in convert.INSIGHTFACE_R50_ALIASES.items()}, convert.INSIGHTFACE_R50_ALIASES)
    assert set(back) == set(P) and all(back[k] is P[k] for k in P)
    # 27.24 M conv parameters, BN statistics not counted
    n_conv = sum(v.size for k, v in P.items() if k.endswith("_weight"))
    assert abs(n_conv - 27.24e6) < 0.02e6


def test_bn_fold_math_and_errors(rfd):
    from rfd_hip import convert
    s, t = convert.bn_fold([2.0], [0.5], [1.0], [4.0 - convert.EPS_DEFAULT])
    assert np.allclose(s, [1.0]) and np.allclose(t, [-0.5])
    g = rfd.Graph(rfd.BACKBONE_R50, 640, 640)
    P = unfolded_ref.make_params(1)

    class Null:
        def set_layer(self, *a): pass
        def set_affine(self, *a): pass

    bad = dict(P)
    del bad["stage2_unit3_conv2_bn_var"]
    with pytest.raises(KeyError):
        convert.import_unfolded(Null(), g, bad)
    bad = dict(P)
    bad["fpn_lat2_weight"] = bad["fpn_lat2_weight"][:, :512]
    with pytest.raises(ValueError):
        convert.import_unfolded(Null(), g, bad)


def test_plan_for_mobilenet(rfd):
    """MobileNet-0.25: every conv (first 3x3, depthwise, pointwise, FPN, SSH) carries a BN to fold, no pre-activation
    affines; shapes are checked by a recording stand-in for the detector."""
    from rfd_hip import convert
    g = rfd.Graph(rfd.BACKBONE_MNET025, 640, 640)
    plan = convert.layer_plan(g)
    assert all(e["affine_bn"] is None for e in plan)
    assert {e["kind"] for e in plan} == {"bn", "head"} and sum(e["kind"] == "head" for e in plan) == 3
    rng = np.random.default_rng(0)
    P = {}
    for L, e in zip(g.layers, plan):
        if e["kind"] == "head":
            st = e["name"][4:]
            for nm, co in (("cls", 4), ("bbox", 8), ("lmk", 20)):
                P["head%s_%s_weight" % (st, nm)] = rng.normal(size=(co, L.cin, 1, 1)).astype(np.float32)
                P["head%s_%s_bias" % (st, nm)] = rng.normal(size=co).astype(np.float32)
        else:
            P[e["name"] + "_weight"] = rng.normal(size=(L.cout, L.cin, L.kh, L.kw)).astype(np.float32)
            for k in ("gamma", "beta", "mean", "var"):
                P["%s_bn_%s" % (e["name"], k)] = rng.uniform(0.5, 1.5, L.cout).astype(np.float32)
    seen = []

    class Rec:
        def set_layer(self, i, w, b):
            L = g.layers[i]
            assert w.shape == (L.cout, L.kh, L.kw, L.cin) and b.shape == (L.cout,)
            seen.append(i)

        def set_affine(self, i, s, t):
            raise AssertionError("no affine expected")

    assert convert.import_unfolded(Rec(), g, P) == set(P) and seen == list(range(g.num_layers))
