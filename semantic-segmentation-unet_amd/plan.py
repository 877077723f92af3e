"""Per-shape step plan: which kernel family runs each layer's forward / data gradient / weight gradient, and in which precision every
tensor of the step is STORED.  Built once per (batch, height, width, mode) from the engine's options and the C library's shape
predicates, then only looked up by `engine.py` -- the schedule itself contains no route or storage predicates any more -- and printable
(`StepPlan.describe()`), so tests assert routes and storage at small sizes as well as at the BASELINE shapes.

Storage rules of the mixed-precision mode (the tests' CPU checker states the same contract as its `Bf16Plan`):
  stage 2 (`bf16_storage`)      a tensor whose ONLY readers are bf16 contractions is stored as bf16: the producer rounds exactly as the
                                readers' operand staging would, so results are bit-identical to fp32 storage -- the BatchNorm output y
                                in front of a bf16 layer, the concat / pooled tensors, every dz of a bf16 layer;
  stage 3 (`bf16_activations`)  the Keras mixed_bfloat16 convention on top: conv outputs r and activation gradients (dx = the dy of the
                                layers below) are stored as bf16 as well, BatchNorm sums are taken before the rounding wherever a conv
                                epilogue supplies them; includes the two ends of the network (first layer's r and dz, the class map's
                                input and input gradient), whose kernels compute in fp32.
"""
import dataclasses
import os
from dataclasses import dataclass, field
from typing import Dict, Optional, Tuple

BASE = 64

F32, BF16 = "f32", "bf16"


@dataclass
class EngineOptions:
    """Every switch of the engine, in one place (constructor argument of `Engine`; `from_env()` reads the UNET_* diagnostics
    variables ONCE at construction -- the C library itself reads no environment)."""
    compute_dtype: str = "fp32"       # "fp32": the reference's arithmetic | "bf16": BASELINE config 4 (bf16 contractions, fp32 master weights)
    conv_route: str = "fused"         # 3x3 forward / data gradient: "fused" = fused Winograd F(2x2,3x3) where it applies | "direct" = implicit GEMM
    fp32_matrix: str = "bf16x6"       # how the fused Winograd and transposed-conv forward / data gradient multiply in fp32 mode: "bf16x6" = fp32 operands as three bf16
    #                                   pieces, six products on the bf16 matrix pipe, fp32 accumulation (fp32-grade; csrc/winograd_x6.hip) where
    #                                   the shape allows | "native" = v_mfma_f32_32x32x2_f32 everywhere (csrc/winograd.hip)
    wgrad_route: str = "fused"        # 3x3 weight gradient: "fused" Winograd | "direct"
    bn_on_load: bool = True           # fp32 fused route: BatchNorm-apply folded into the consumer's weights (13 layers)
    fuse_bn_stats: bool = True        # BatchNorm sums from conv / data-gradient epilogues instead of reduction passes
    fuse_pool: bool = True            # BatchNorm apply + max pool in one pass; pool backward inside the BatchNorm backward
    merge_bn_finalize: bool = False   # statistics finalize inside the BatchNorm-apply launch (unet_bn_finalize_apply_any): one dependent launch fewer per layer.
    #                                   OFF: measured slower in same-box A/Bs (profiles/r06_small_launch_ab.txt: bf16 +0.5 ms, fp32 +0.2 ms per step)
    bf16_storage: bool = True         # stage 2 (see module docstring)
    bf16_activations: bool = True     # stage 3
    overlap_wgrad: bool = True        # weight gradients on a side stream
    max_workgroups: Optional[int] = None     # cap on every persistent grid (the `_wg` entry points of include/unet_hip.h: Winograd forward / data /
    #                                   weight gradient, bf16 3x3 kernels, transposed-conv forward and weight gradients): None / 0 = one
    #                                   workgroup per CU, n = at most n workgroups (224 leaves ~4 CUs per XCD to a collective's kernels;
    #                                   measured on one GPU with a stand-in collective it loses, so nothing sets it -- parallel.py)

    @staticmethod
    def from_env(env=None):
        env = os.environ if env is None else env
        o = EngineOptions()
        o.compute_dtype = env.get("UNET_COMPUTE_DTYPE", o.compute_dtype)
        o.conv_route = env.get("UNET_CONV_ROUTE", o.conv_route)
        o.wgrad_route = env.get("UNET_WGRAD_ROUTE", o.wgrad_route)
        o.fp32_matrix = env.get("UNET_FP32_MATRIX", o.fp32_matrix)
        for name, var in (("bn_on_load", "UNET_BN_ON_LOAD"), ("fuse_bn_stats", "UNET_FUSE_BN_STATS"), ("fuse_pool", "UNET_FUSE_POOL"),
                          ("merge_bn_finalize", "UNET_MERGE_BN_FINALIZE"), ("bf16_storage", "UNET_BF16_STORAGE"), ("bf16_activations", "UNET_BF16_ACTIVATIONS"),
                          ("overlap_wgrad", "UNET_OVERLAP_WGRAD")):
            if var in env:
                setattr(o, name, env[var] != "0")
        if "UNET_MAX_WORKGROUPS" in env:
            o.max_workgroups = int(env["UNET_MAX_WORKGROUPS"])
        if o.compute_dtype not in ("fp32", "bf16"):
            raise ValueError("compute_dtype must be 'fp32' or 'bf16'")
        if o.fp32_matrix not in ("bf16x6", "native"):
            raise ValueError("fp32_matrix must be 'bf16x6' or 'native'")
        return o

    def key(self):
        return dataclasses.astuple(self)

    @property
    def cap(self):
        """the `max_workgroups` argument the C entry points get (0 = one workgroup per CU)"""
        return int(self.max_workgroups or 0)


# layer -> (producer, first, last, parts): the layer's input channels [first/parts, last/parts) of its Cin are EXACTLY the BatchNorm
# output of `producer` and feed nothing else, so the layer's data gradient over that range is the producer's dy
PRODUCER = {"bott_b": ("bott_a", 0, 1, 1)}
for _l in (1, 2, 3, 4):
    PRODUCER["conv_%db" % _l] = ("conv_%da" % _l, 0, 1, 1)
    PRODUCER["dec_%db" % _l] = ("dec_%da" % _l, 0, 1, 1)
    PRODUCER["dec_%da" % _l] = ("up_%d" % _l, 1, 2, 2)          # concat [skip, upsampled] (UNet/model.py:55-58): upper half
for _l in (1, 2, 3):
    PRODUCER["up_%d" % _l] = ("dec_%db" % (_l + 1), 0, 1, 1)   # (up_4's input went through the dropout: no direct producer)

# layer -> the ONE layer that reads its BatchNorm output through a contraction (None: several readers / a pooled or dropped tensor)
CONSUMER = {"bott_a": "bott_b", "bott_b": "up_4", "dec_1b": "logits"}
for _l in (1, 2, 3, 4):
    CONSUMER["conv_%da" % _l] = "conv_%db" % _l
    CONSUMER["dec_%da" % _l] = "dec_%db" % _l
    CONSUMER["up_%d" % _l] = "dec_%da" % _l
for _l in (2, 3, 4):
    CONSUMER["dec_%db" % _l] = "up_%d" % (_l - 1)


@dataclass
class LayerPlan:
    name: str
    kind: str
    cin: int
    cout: int
    hi: int
    wi: int
    ho: int
    wo: int
    fwd: str = ""                  # conv3: bf16 | winograd | mfma | direct;  deconv: convt_bf16 | convt_x6 | convt_stream | convt_igemm;  conv1: conv1x1
    dgrad: str = ""                # same families; "none" = not computed in a training step (first layer)
    wgrad: str = ""
    fwd_x6: bool = False           # fwd == "winograd": the BF16x6 kernel (fp32-grade products on the bf16 matrix pipe) instead of the fp32-MFMA one
    dgrad_x6: bool = False         # the same for the data gradient
    x_on_load: bool = False        # the layer reads its producer's conv output through BatchNorm-apply on load (fp32 Winograd route)
    defer_y: bool = False          # ... and this layer's own BatchNorm output is not materialised (its consumer applies it on load)
    r: str = F32                   # storage of the conv output (post-ReLU, pre-BatchNorm)
    y: Optional[str] = F32         # storage of the BatchNorm output (None: deferred).  conv_Nb: the skip half of the concat buffer
    dz: str = F32                  # storage of the BatchNorm-backward output
    dx: Optional[str] = F32        # storage of the data gradient (None: not computed)
    fwd_stats: bool = False        # BatchNorm sums of the output come from the forward kernel's epilogue
    sums_from_dgrad: bool = False  # BatchNorm-backward sums come from the consumer's data-gradient epilogue
    leaves_sums_for: Optional[Tuple[str, int, int]] = None     # (producer, c0, c1): this layer's data gradient leaves them


@dataclass
class StepPlan:
    n: int
    h: int
    w: int
    training: bool
    want_grad: bool
    options: Tuple
    layer: Dict[str, LayerPlan] = field(default_factory=dict)
    cat: Dict[int, str] = field(default_factory=dict)          # level -> storage of the concat buffer [skip, upsampled] and of the pooled tensor
    fuse_pool: Dict[int, bool] = field(default_factory=dict)   # level -> BatchNorm apply + pool in one pass (forward) / pool gradient inside BatchNorm backward
    lvl4_grad: str = F32                                       # storage of dec_4a's / bott_a's data gradients (added into each other)

    def describe(self):
        rows = ["plan n=%d h=%d w=%d training=%s want_grad=%s" % (self.n, self.h, self.w, self.training, self.want_grad)]
        for p in self.layer.values():
            rows.append("%-8s %-6s %4d->%-4d @%dx%d  fwd=%-12s dgrad=%-11s wgrad=%-10s r=%s y=%s dz=%s dx=%s%s%s%s%s" % (
                p.name, p.kind, p.cin, p.cout, p.ho, p.wo, p.fwd + ("/x6" if p.fwd_x6 else ""), p.dgrad + ("/x6" if p.dgrad_x6 else ""), p.wgrad, p.r, p.y, p.dz, p.dx,
                " on_load" if p.x_on_load else "", " defer_y" if p.defer_y else "", " fwd_stats" if p.fwd_stats else "",
                " sums<-" + CONSUMER.get(p.name, "?") if p.sums_from_dgrad else ""))
        rows.append("concat/pool storage: " + " ".join("L%d=%s%s" % (l, d, "(fused)" if self.fuse_pool[l] else "") for l, d in self.cat.items()))
        return "\n".join(rows)

    def rounding_points(self):
        """-> every point where this plan rounds to bf16, as sets of layer names (the tests' checker states the same contract on its
        side -- `Bf16Plan` -- and the two are compared field by field)."""
        L = self.layer
        return dict(
            contract=frozenset(n for n, p in L.items() if p.fwd in ("bf16", "convt_bf16") and p.wgrad in ("bf16", "convt_bf16")),
            r_bf16=frozenset(n for n, p in L.items() if p.r == BF16),
            # (a bf16 y in front of a bf16 contraction is the operand rounding itself; it is a rounding point of its own only where the
            # one reader computes in fp32)
            y_bf16=frozenset(n for n, p in L.items() if p.y == BF16 and n in CONSUMER and L[CONSUMER[n]].fwd not in ("bf16", "convt_bf16")),
            dz_bf16=frozenset(n for n, p in L.items() if p.dz == BF16),
            dx_bf16=frozenset(n for n, p in L.items() if p.dx == BF16),
            sums_from_dgrad=frozenset(n for n, p in L.items() if p.sums_from_dgrad),
            lvl4_accumulate_bf16=self.lvl4_grad == BF16)


def layer_table(number_channels, number_classes):
    """(name, kind, Cin, Cout) in Keras layer-creation order (reference UNet/model.py:85-136)."""
    C, K, B = number_channels, number_classes, BASE
    return [
        ("conv_1a", "conv3", C, B), ("conv_1b", "conv3", B, B),
        ("conv_2a", "conv3", B, 2 * B), ("conv_2b", "conv3", 2 * B, 2 * B),
        ("conv_3a", "conv3", 2 * B, 4 * B), ("conv_3b", "conv3", 4 * B, 4 * B),
        ("conv_4a", "conv3", 4 * B, 8 * B), ("conv_4b", "conv3", 8 * B, 8 * B),
        ("bott_a", "conv3", 8 * B, 16 * B), ("bott_b", "conv3", 16 * B, 16 * B),
        ("up_4", "deconv", 16 * B, 8 * B), ("dec_4a", "conv3", 16 * B, 8 * B), ("dec_4b", "conv3", 8 * B, 8 * B),
        ("up_3", "deconv", 8 * B, 4 * B), ("dec_3a", "conv3", 8 * B, 4 * B), ("dec_3b", "conv3", 4 * B, 4 * B),
        ("up_2", "deconv", 4 * B, 2 * B), ("dec_2a", "conv3", 4 * B, 2 * B), ("dec_2b", "conv3", 2 * B, 2 * B),
        ("up_1", "deconv", 2 * B, B), ("dec_1a", "conv3", 2 * B, B), ("dec_1b", "conv3", B, B),
        ("logits", "conv1", B, K),
    ]


def _level(name):
    """encoder / decoder level 1..4 of a layer (5 = bottleneck), by which its spatial size is the image's >> (level - 1)"""
    if name.startswith("bott"):
        return 5
    if name == "logits":
        return 1
    return int(name.split("_")[1][0])


def build_plan(opt, number_channels, number_classes, n, h, w, training, want_grad, L):
    """opt: EngineOptions; L: the C library binding (shape predicates only -- no device work)."""
    pl = StepPlan(n, h, w, bool(training), bool(want_grad), opt.key())
    bf = opt.compute_dtype == "bf16"
    train = bool(training)
    two_gib = 2 ** 31
    for name, kind, cin, cout in layer_table(number_channels, number_classes):
        lvl = _level(name)
        if kind == "deconv":
            hi, wi = h >> lvl, w >> lvl
            ho, wo = 2 * hi, 2 * wi
        else:
            hi = ho = h >> (lvl - 1)
            wi = wo = w >> (lvl - 1)
        pl.layer[name] = LayerPlan(name, kind, cin, cout, hi, wi, ho, wo)

    # ---- kernel families ---------------------------------------------------------------------------------------------------------------
    def wino_ok(p, dgrad=False):
        k, nn = (p.cout, p.cin) if dgrad else (p.cin, p.cout)
        return opt.conv_route == "fused" and p.ho % 2 == 0 and p.wo % 2 == 0 and k % 8 == 0 and nn % 64 == 0

    for p in pl.layer.values():
        if p.kind == "conv3":
            # the bf16 kernels address their tensors with 32-bit buffer offsets: every operand (leading dimension <= max(Cin, Cout): a
            # concat input IS the layer's Cin) must stay below 2 GiB, larger problems fall back to the fp32 kernels
            small = n * p.ho * p.wo * max(p.cin, p.cout) * 4 < two_gib
            b16 = (bf and small and L.unet_conv3x3_bf16_supported(n, p.ho, p.wo, p.cin, p.cout) == 1
                   and L.unet_conv3x3_bf16_supported(n, p.ho, p.wo, p.cout, p.cin) == 1)
            p.fwd = "bf16" if b16 else "winograd" if wino_ok(p) else "mfma" if L.unet_conv3x3_mfma_supported(p.cin, p.cout) else "direct"
            p.dgrad = "bf16" if b16 else "winograd" if wino_ok(p, True) else "mfma" if L.unet_conv3x3_mfma_supported(p.cout, p.cin) else "direct"
            if opt.fp32_matrix == "bf16x6":
                p.fwd_x6 = p.fwd == "winograd" and L.unet_winograd_x6_supported(n, p.ho, p.wo, p.cin, p.cout) == 1
                p.dgrad_x6 = p.dgrad == "winograd" and L.unet_winograd_x6_supported(n, p.ho, p.wo, p.cout, p.cin) == 1
            if bf and small and L.unet_conv3x3_wgrad_bf16_supported(n, p.ho, p.wo, p.cin, p.cout) == 1:
                p.wgrad = "bf16"
            elif opt.wgrad_route == "fused" and L.unet_winograd_wgrad_fused_supported(n, p.ho, p.wo, p.cin, p.cout) == 1:
                p.wgrad = "winograd"
            elif L.unet_conv3x3_mfma_supported(p.cin, p.cout) and p.cin % 64 == 0:
                p.wgrad = "mfma"
            else:
                p.wgrad = "direct"
        elif p.kind == "deconv":
            b16 = (bf and n * p.hi * p.wi * 4 * p.cout * 4 < two_gib and n * p.hi * p.wi * p.cin * 4 < two_gib
                   and L.unet_convT2x2_bf16_supported(n, p.hi, p.wi, p.cin, p.cout) == 1)
            x6 = (not b16) and opt.fp32_matrix == "bf16x6" and L.unet_convT2x2_x6_supported(n, p.hi, p.wi, p.cin, p.cout) == 1
            p.fwd = ("convt_bf16" if b16 else "convt_x6" if x6 else
                     "convt_stream" if L.unet_convT2x2_fwd_stream_supported(n, p.hi, p.wi, p.cin, p.cout) == 1 else "convt_igemm")
            p.dgrad = "convt_bf16" if b16 else "convt_x6" if x6 else "convt_igemm"
            p.wgrad = ("convt_bf16" if (b16 and L.unet_convT2x2_wgrad_bf16_supported(n, p.hi, p.wi, p.cin, p.cout) == 1) else
                       "convt_x6" if x6 else "convt")
        else:
            p.fwd = p.dgrad = p.wgrad = "conv1x1"
    first = pl.layer["conv_1a"]
    if train:
        first.dgrad, first.dx = "none", None                     # nothing below the first layer wants a gradient (the ERF probe runs in eval mode)

    def allbf(name):
        """forward, data gradient and weight gradient of the layer all run on the bf16 kernels: every read of its input, output gradient
        and (data gradient) weights rounds to bf16 the same way"""
        p = pl.layer[name]
        return p.fwd in ("bf16", "convt_bf16") and p.wgrad in ("bf16", "convt_bf16")

    # ---- fused BatchNorm statistics ------------------------------------------------------------------------------------------------------
    for p in pl.layer.values():
        if not (train and opt.fuse_bn_stats):
            continue
        if p.fwd == "bf16":
            p.fwd_stats = L.unet_conv3x3_bf16_stats_rows(n, p.ho, p.wo, p.cin, p.cout) > 0
        elif p.fwd == "winograd":
            p.fwd_stats = L.unet_conv3x3_fwd_winograd_fused_stats_rows_wg(n, p.ho, p.wo, p.cin, p.cout, opt.cap) > 0
        elif p.fwd == "direct":
            p.fwd_stats = L.unet_conv3x3_fwd_direct_stats_rows(n, p.ho, p.wo, p.cin, p.cout) > 0
        elif p.fwd == "convt_bf16":
            p.fwd_stats = L.unet_convT2x2_bf16_stats_rows(n, p.hi, p.wi, p.cin, p.cout, 0) > 0
        elif p.fwd == "convt_x6":
            p.fwd_stats = L.unet_convT2x2_x6_stats_rows(n, p.hi, p.wi, p.cin, p.cout) > 0
        elif p.fwd == "convt_stream":
            p.fwd_stats = L.unet_convT2x2_fwd_stream_stats_rows_wg(n, p.hi, p.wi, p.cin, p.cout, opt.cap) > 0

    # ---- storage -------------------------------------------------------------------------------------------------------------------------
    st2 = bf and opt.bf16_storage
    st3 = st2 and opt.bf16_activations
    for p in pl.layer.values():
        # dz: read by the layer's own data / weight gradient kernels only
        if train and st2 and allbf(p.name) and (p.dgrad in ("bf16", "convt_bf16")):
            p.dz = BF16
        # r: what BatchNorm (forward apply, backward) reads; bf16 when the BatchNorm backward takes the bf16-capable entry point (dz bf16)
        if train and st3 and p.fwd_stats and p.dz == BF16:
            p.r = BF16
    if train and st3 and first.fwd == "direct" and first.fwd_stats and first.cout % 8 == 0:
        first.r = first.dz = BF16                                # the fp32 stencil kernels of the first layer write / read either storage
    # y: bf16 in front of a layer whose forward and weight gradient round it anyway; the class map's input in a training step (stage 3)
    for name, p in pl.layer.items():
        cons = CONSUMER.get(name)
        if cons is not None and cons != "logits" and st2 and allbf(cons):
            p.y = BF16
    if train and want_grad and st3:
        pl.layer["dec_1b"].y = BF16
    # concat [skip, upsampled] + pooled tensors of a level: read by dec_Na and the next level's first conv
    for lvl in (1, 2, 3, 4):
        nxt = "conv_%da" % (lvl + 1) if lvl < 4 else "bott_a"
        pl.fuse_pool[lvl] = opt.fuse_pool and not (lvl == 4 and train)          # level 4 drops out between BatchNorm and pool (UNet/model.py:105-107)
        c16 = st2 and opt.fuse_pool and (lvl < 4 or st3) and allbf("dec_%da" % lvl) and allbf(nxt)
        pl.cat[lvl] = BF16 if c16 else F32
        pl.layer["conv_%db" % lvl].y = pl.cat[lvl]
        pl.layer["up_%d" % lvl].y = pl.cat[lvl]

    # ---- BatchNorm-apply on load (fp32 fused Winograd route): producer -> consumer pairs whose intermediate is never materialised ---------
    def can_defer(a, b):
        pb = pl.layer[b]
        return (opt.bn_on_load and not bf and opt.wgrad_route == "fused" and pb.fwd == "winograd" and pb.wgrad == "winograd"
                and pl.layer[a].y == F32)
    for lvl in (1, 2, 3, 4):
        for a, b in (("conv_%da" % lvl, "conv_%db" % lvl), ("dec_%da" % lvl, "dec_%db" % lvl), ("up_%d" % lvl, "dec_%da" % lvl)):
            if can_defer(a, b):
                pl.layer[a].defer_y, pl.layer[a].y, pl.layer[b].x_on_load = True, None, True
    if can_defer("bott_a", "bott_b"):
        pl.layer["bott_a"].defer_y, pl.layer["bott_a"].y, pl.layer["bott_b"].x_on_load = True, None, True

    # ---- backward: sums from data-gradient epilogues, storage of the data gradients ----------------------------------------------------------
    if train:
        for name, p in pl.layer.items():
            prod = PRODUCER.get(name)
            if prod is None or not opt.fuse_bn_stats or p.dgrad == "none":
                continue
            pp = pl.layer[prod[0]]
            ok = False
            if p.dgrad == "bf16":
                ok = L.unet_conv3x3_bf16_stats_rows(n, p.ho, p.wo, p.cout, p.cin) > 0
            elif p.dgrad == "winograd":
                ok = pp.r == F32 and L.unet_conv3x3_fwd_winograd_fused_stats_rows_wg(n, p.ho, p.wo, p.cout, p.cin, opt.cap) > 0
            elif p.dgrad == "convt_bf16":
                ok = L.unet_convT2x2_bf16_stats_rows(n, p.hi, p.wi, p.cin, p.cout, 1) > 0
            elif p.dgrad == "convt_x6":
                ok = pp.r == F32 and L.unet_convT2x2_x6_bnbwd_rows(n, p.hi, p.wi, p.cin, p.cout) > 0
            if ok:
                c0, c1 = prod[1] * (p.cin // prod[3]), prod[2] * (p.cin // prod[3])
                p.leaves_sums_for = (prod[0], c0, c1)
                pp.sums_from_dgrad = True

        # a data gradient may be WRITTEN as bf16 when every reader of it is a BatchNorm backward that takes bf16 dy (its dz is bf16)
        takes16 = lambda nm: pl.layer[nm].dz == BF16
        d4 = (st3 and pl.layer["dec_4a"].dgrad == "bf16" and pl.layer["bott_a"].dgrad == "bf16" and pl.cat[4] == BF16)
        pl.lvl4_grad = BF16 if d4 else F32
        for name, p in pl.layer.items():
            if not st3 or p.dx is None:
                continue
            ok = False
            if p.kind == "conv1":                                  # class map: its input gradient is dec_1b's dy
                ok = p.cin % 8 == 0 and takes16("dec_1b")
            elif p.kind == "deconv":
                ok = p.dgrad == "convt_bf16" and p.leaves_sums_for is not None and takes16(p.leaves_sums_for[0])
            elif p.dgrad == "bf16":                                # (the fp32 kernels of a size fallback write fp32)
                if name.startswith("dec_") and name.endswith("a"):     # [skip, upsampled]: conv_Nb (through the fused pool path) and up_N
                    lvl = _level(name)
                    if lvl == 4:     # skip half -> pool-backward accumulate + dropout kernels (either storage), then conv_4b's BatchNorm backward
                        ok = d4 and takes16("up_4")
                    else:
                        ok = opt.fuse_pool and takes16("conv_%db" % lvl) and takes16("up_%d" % lvl)
                elif name.startswith("conv_") and name.endswith("a"):  # pooled gradient of the level above, formed inside its BatchNorm backward
                    lvl = _level(name)
                    ok = lvl >= 2 and opt.fuse_pool and takes16("conv_%db" % (lvl - 1))
                elif name == "bott_a":                                 # read by the pool-backward kernel, added into dec_4a's skip gradient
                    ok = d4
                else:
                    prod = PRODUCER.get(name)
                    ok = prod is not None and prod[3] == 1 and takes16(prod[0])
            if ok:
                p.dx = BF16
    else:
        for p in pl.layer.values():
            p.dz = F32
    return pl
