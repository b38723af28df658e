"""Execution switches of the Python host as EXPLICIT state (the twin of the C side's bmv_tuning_*, csrc/tuning.hip).

Every switch is declared here with its default and what it does; `get(name)` is what the networks read.  Environment
variables of the same names are applied ONCE, here, at import (in the open, as `_lib.py` does for the library's
tuning switches) -- nothing else in the package reads `os.environ` for behaviour.  They select between measured
alternatives of the SAME HIP path (tests compare them; A/B logs under profiles/); none of them is a CPU fallback.

    from boostmvsnerfs_amd import switches
    switches.set("BMV_BOOST_STREAMS", 0)        # before the network is constructed
    with switches.override(BMV_CNN="torch"):    # tests: the torch modules instead of the convolution engine
        ...
"""
import contextlib
import os

# name -> (default, meaning)
DECLARED = {
    # networks/enerf/network.py
    "BMV_OVERLAP": (2, "level-0 chain under FeatureNet's top-down path: 0 one stream, 1 whole chain forked, 2 what follows the sweep"),
    "BMV_OVERLAP_EAGER": (0, "1 = fork outside HIP-graph capture too"),
    "BMV_DEFER_F0": (0, "1 = FeatureNet's full-resolution map (only the renderer reads it) on the side stream beside the level-1 chain"),
    "BMV_LOOKUP_RECORDS": (1, "FeatureNet writes the renderer's image lookup records in its epilogue"),
    "BMV_VOLUME_RECORDS": (1, "the regularisers' heads write the renderer's volume records"),
    "BMV_FRAME_SETUP": (1, "cameras, projections and level-0 hypotheses of a frame in one launch"),
    "BMV_DEPTH_MAPS_TABLE": (1, "depth_regress writes the frame's depth / std maps through the pointer table itself"),
    "BMV_SIDE_PRIO": (0, "priority of the side stream"),
    "BMV_CHECK_NAN": (0, "1 = raise on NaN in the rendered rgb (debugging)"),
    # networks/boost_enerf/network.py
    "BMV_BOOST_SIDE_SETUP": (1, "K-volume camera set-up on a side stream under FeatureNet"),
    "BMV_BOOST_STREAMS": (1, "the K cost-volume chains on K HIP streams"),
    "BMV_BOOST_BATCHED": (0, "the K cost volumes as one batch through the regularisers (measured slower: opt-in)"),
    # networks/enerf/cnn.py, convnet.py, conv_train.py
    "BMV_CNN": ("engine", "'torch' = the torch modules instead of the convolution engine (tests compare the two)"),
    "BMV_BN": ("hip", "'torch' = torch's batch norm in training mode"),
    "BMV_FPN_FUSE": (1, "fused FPN top-down + smooth0"),
    "BMV_FPN_S": (1, "... on the bf16 matrix cores with three-piece fp32 operands and lat0 folded into smooth0's weights (csrc/fpn_s.hip; the default since round 6); 0 = the fp32 kernel of csrc/conv.hip"),
    "BMV_CONV0_FUSE": (1, "fused first FeatureNet block"),
    "BMV_BOOST_OVERLAP": (0, "1 = K-volume ENeRF (boost_enerf) inference with the K level-0 chains on their streams UNDER FeatureNet's top-down path (which only the level-1 sweeps and the renderer need) instead of behind it.  Measured round 6: 2.235 -> 2.21 ms at 6 x 480 x 736, K = 4 -- and hipGraphLaunch of that topology (every volume stream joins the main stream twice) SEGFAULTS on ROCm 7.2 in a process that captured other graphs before (tests/test_gpu_framegraph.py in file order): off by default"),
    "BMV_CONV2D_S": (1, "FeatureNet's conv1.0 / conv1.1 / conv2.0 (5x5 stride 2, 3x3) on the bf16 matrix cores, three-piece fp32 operands (csrc/conv2d_s.hip; round 6); 0 = the fp32 engine of csrc/conv.hip"),
    "BMV_CONV2D_S_REC": (1, "... with the maps between those layers as split records (three bf16 pieces per value, written by the producing layer's epilogue, staged by LDS-DMA: convnet.SplitRecords; bit-identical results); 0 = planar fp32 maps, every consuming wave splits"),
    "BMV_CONV0_S": (1, "... with its second layer on the bf16 matrix cores, three-piece fp32 operands (csrc/fpn_s.hip conv0_s_kernel; the default since round 6); 0 = the fp32 kernel of csrc/conv.hip"),
    "BMV_TOP_FUSE": (1, "fused conv2 tail + top layer"),
    "BMV_CONV_SPLIT": ("0", "split-bf16 first layers / heads: '0' fp32 engine, 'auto' / '3' three pieces, '2' two pieces (opt-in experiment)"),
    "BMV_QUAD_VOLUME": (1, "the inference sweep hands the regulariser's first layer its cost volume as 16-byte quad records"),
    "BMV_QUAD_S0": (1, "... and the first layer's output (stride-2 layer input, conv11's skip) as quad records too"),
    "BMV_CONV_C4": (1, "<= 9-output-channel layers on v_mfma_f32_4x4x1_16b_f32 (csrc/conv_c4.hip)"),
    "BMV_CONV_C4S": (1, "... and, where their input arrives as quad records, as bf16 MFMAs on three-piece fp32 operands at fp32 accuracy (csrc/conv_c4s.hip; the default since round 6); 0 = the fp32 blocks"),
    "BMV_TRAIN_CONV": ("engine", "'torch' = MIOpen for the training convolutions"),
    "BMV_TRAIN_DGRAD5": (1, "stride-2 5x5 data gradients as one 3x3 engine convolution"),
    # ops.py
    "BMV_SWEEP_BWD": ("cl", "'cl' = the channel-last sweep backward (csrc/sweep_bwd_cl.hip) where it applies, else the planar kernel"),
    # networks/mvsnerf/network.py
    "BMV_MVS_MLP_TRAIN": ("hip", "'torch' = MVSNeRF's training MLP as torch ops (tests compare the two)"),
    # autograph.py
    "BMV_AUTOGRAPH": (1, "self-capturing forward (HIP graph replay)"),
    "BMV_AUTOGRAPH_DEFER": (1, "large inputs / outputs through a pointer table"),
    "BMV_AUTOGRAPH_RING": (1, "the table is fed by the frame's first node from a host ring"),
    "BMV_AUTOGRAPH_MAX": (4, "captured graphs kept per network"),
}

VALUES = {}


def _parse(name, raw):
    default = DECLARED[name][0]
    raw = raw.strip()
    if isinstance(default, int):
        try:
            return int(raw, 0)
        except ValueError:
            try:
                return int(raw)           # "08" and the like, as `_lib.load` reads the library's tuning switches
            except ValueError:
                pass
            low = raw.lower()
            if low in ("true", "on", "yes"):
                return 1
            if low in ("false", "off", "no", ""):
                return 0
            raise ValueError(f"environment variable {name}={raw!r}: expected an integer (or true / false)") from None
    return raw


def get(name):
    if name not in DECLARED:
        raise KeyError(f"unknown switch {name!r} (boostmvsnerfs_amd/switches.py declares them)")
    return VALUES.get(name, DECLARED[name][0])


def on(name):
    return get(name) == 1


def set(name, value):  # noqa: A001 (the C side's verb: bmv_tuning_set)
    if name not in DECLARED:
        raise KeyError(f"unknown switch {name!r} (boostmvsnerfs_amd/switches.py declares them)")
    default = DECLARED[name][0]
    VALUES[name] = int(value) if isinstance(default, int) else str(value)


def clear(name=None):
    if name is None:
        VALUES.clear()
    else:
        VALUES.pop(name, None)


@contextlib.contextmanager
def override(**kw):
    saved = {k: VALUES.get(k, None) for k in kw}
    try:
        for k, v in kw.items():
            set(k, v)
        yield
    finally:
        for k, v in saved.items():
            if v is None:
                VALUES.pop(k, None)
            else:
                VALUES[k] = v


def describe():
    return {k: {"value": get(k), "default": d, "what": w} for k, (d, w) in DECLARED.items()}


def apply_environment(environ=None):
    """Environment variables named like a declared switch become its value (once, at import)."""
    environ = os.environ if environ is None else environ
    for name in DECLARED:
        if name in environ:
            VALUES[name] = _parse(name, environ[name])


apply_environment()
