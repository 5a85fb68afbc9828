"""Parameter table of the CtRL-Sim encoder/decoder and the build-owned weight generator (G0).

The table mirrors the Lightning state_dict of `models/ctrl_sim.py:CtRLSim` (keys `encoder.*`,
`decoder.*`) so that a real checkpoint's `state_dict` can be loaded with
`from_state_dict`.  No trained checkpoint exists in this environment, so `generate()` produces
synthetic weights with the distribution of `utils/train_utils.py:13-78` (Xavier-uniform Linear
weights, N(0,0.02) embeddings, in_proj U(+-sqrt(6/(2D)))) from a counter-based generator keyed
by (seed, parameter name, flat index).  Unlike `weight_init`, biases and LayerNorm affine
parameters are drawn small-but-nonzero so that every bias/affine path is exercised by the
parity tests (a trained checkpoint has them nonzero too).
"""
from __future__ import annotations

import numpy as np

from .spec import Dims

_MASK = (1 << 64) - 1


def _fnv1a64(s: str) -> int:
    h = 0xCBF29CE484222325
    for b in s.encode():
        h = ((h ^ b) * 0x100000001B3) & _MASK
    return h


def splitmix64(x: np.ndarray) -> np.ndarray:
    """Vectorised splitmix64 finaliser on uint64 arrays (wrap-around arithmetic)."""
    with np.errstate(over="ignore"):
        x = (x + np.uint64(0x9E3779B97F4A7C15)).astype(np.uint64)
        z = x
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform01(key: int, n: int, offset: int = 0) -> np.ndarray:
    """n float64 uniforms in (0,1) with 24-bit resolution: u = (top24 + 0.5) / 2^24."""
    idx = np.arange(offset, offset + n, dtype=np.uint64)
    with np.errstate(over="ignore"):
        z = splitmix64(np.uint64(key & _MASK) + idx * np.uint64(0xD1342543DE82EF95))
    return ((z >> np.uint64(40)).astype(np.float64) + 0.5) / float(1 << 24)


def param_table(dims: Dims):
    """[(name, shape, kind, arg)] in state_dict order.  kind: 'xavier' (arg unused), 'uniform' (bound),
    'normal' (std), 'bias', 'ln_w', 'ln_b'."""
    D, F, H = dims.D, dims.F, dims.H
    t = []

    def linear(name, fin, fout):
        t.append((f"{name}.weight", (fout, fin), "xavier", None))
        t.append((f"{name}.bias", (fout,), "bias", None))

    def ln(name, d=D):
        t.append((f"{name}.weight", (d,), "ln_w", None))
        t.append((f"{name}.bias", (d,), "ln_b", None))

    def mlp(name, fin, hid, fout):  # utils/layers.py:6-19  Linear-LN-ReLU-Linear
        linear(f"{name}.mlp.0", fin, hid)
        ln(f"{name}.mlp.1", hid)
        linear(f"{name}.mlp.3", hid, fout)

    def mha(name):
        t.append((f"{name}.in_proj_weight", (3 * D, D), "uniform", (6.0 / (2 * D)) ** 0.5))
        t.append((f"{name}.in_proj_bias", (3 * D,), "bias", None))
        linear(f"{name}.out_proj", D, D)

    # modules/map_encoder.py:16-26
    t.append(("encoder.map_encoder.map_seeds", (1, 1, D), "uniform", (6.0 / (D + D)) ** 0.5))
    mlp("encoder.map_encoder.road_pts_encoder", 3, D, D)
    mha("encoder.map_encoder.road_pts_attn_layer")
    ln("encoder.map_encoder.norm1")
    ln("encoder.map_encoder.norm2")
    mlp("encoder.map_encoder.map_feats", D, D, D)
    mlp("encoder.map_encoder.road_type_encoder", 8, D, D)
    mlp("encoder.map_encoder.road_road_type_encoder", 2 * D, D, D)
    # modules/encoder.py:21-46
    mlp("encoder.embed_state", dims.STATE, D, D)
    mlp("encoder.embed_goal", dims.GOAL, D, D)
    linear("encoder.embed_state_goal", 2 * D, D)
    t.append(("encoder.embed_action.weight", (dims.V, D), "normal", 0.02))
    for c in ("goal", "veh", "road"):
        if getattr(dims, "VARIANT", 0) == 3:      # decision transformer: nn.Linear(1, D) per component (encoder.py:27-30)
            linear(f"encoder.embed_rtg_{c}", 1, D)
        else:
            t.append((f"encoder.embed_rtg_{c}.weight", (dims.R, D), "normal", 0.02))
    linear("encoder.embed_rtg", D * dims.C, D)
    t.append(("encoder.embed_timestep.weight", (dims.MAXT, D), "normal", 0.02))
    t.append(("encoder.embed_agent_id.weight", (dims.A, D), "normal", 0.02))
    ln("encoder.embed_ln")
    for i in range(dims.NE):
        p = f"encoder.transformer_encoder.layers.{i}"
        mha(f"{p}.self_attn")
        linear(f"{p}.linear1", D, F)
        linear(f"{p}.linear2", F, D)
        ln(f"{p}.norm1")
        ln(f"{p}.norm2")
    # modules/decoder.py:16-27
    for i in range(dims.ND):
        p = f"decoder.transformer_decoder.layers.{i}"
        mha(f"{p}.self_attn")
        mha(f"{p}.multihead_attn")
        linear(f"{p}.linear1", D, F)
        linear(f"{p}.linear2", F, D)
        ln(f"{p}.norm1")
        ln(f"{p}.norm2")
        ln(f"{p}.norm3")
    mlp("decoder.predict_action", D, D, dims.V)
    if not getattr(dims, "VARIANT", 0):       # the IL / Trajeglish models have neither head (cfgs/model/{il,trajeglish}.yaml)
        mlp("decoder.predict_rtg", D, D, dims.R * dims.C)
        mlp("decoder.predict_future_states", D, D, dims.FUT)
    return t


def generate(dims: Dims, seed: int = 0) -> dict:
    """name -> float32 ndarray.  Deterministic function of (dims, seed) only."""
    out = {}
    for name, shape, kind, arg in param_table(dims):
        n = int(np.prod(shape))
        key = _fnv1a64(name) ^ ((seed * 0x9E3779B97F4A7C15) & _MASK)
        if kind in ("xavier", "uniform"):
            bound = arg if kind == "uniform" else (6.0 / (shape[0] + shape[1])) ** 0.5
            v = (uniform01(key, n) * 2.0 - 1.0) * bound
        elif kind == "normal":
            u1 = uniform01(key, n)
            u2 = uniform01(key ^ 0x5851F42D4C957F2D, n)
            v = np.sqrt(-2.0 * np.log(u1)) * np.cos(2.0 * np.pi * u2) * arg
        elif kind == "bias":
            v = (uniform01(key, n) * 2.0 - 1.0) * 0.05
        elif kind == "ln_w":
            v = 1.0 + (uniform01(key, n) * 2.0 - 1.0) * 0.1
        elif kind == "ln_b":
            v = (uniform01(key, n) * 2.0 - 1.0) * 0.05
        else:  # pragma: no cover
            raise ValueError(kind)
        out[name] = v.astype(np.float32).reshape(shape)
    return out


def generate_trained_like(dims: Dims, seed: int = 0, logit_gain: float = 15.0) -> dict:
    """Synthetic weights with the statistics of a TRAINED checkpoint instead of `weight_init`'s (round-4 review: eval_sim.py:52 loads a
    trained model; every earlier fixture had |logit| <= 2.6, LayerNorm gains of 1 +- 0.1 and logits that hardly depended on the scene).
    Deterministic function of (dims, seed):

      * LayerNorm gains log-uniform in [0.5, 2], LayerNorm biases U(+-0.15), Linear biases U(+-0.1);
      * every 2-D Linear / in_proj weight of `generate()` scaled per output row AND per input column, both log-uniform in [0.7, 1.4]
        (entries of one matrix span > 10^3 in magnitude together with the uniform draw);
      * the query and key rows of every in_proj x 2 (scores x 4: peaked attention, as trained attention is — a random-init softmax
        averages its keys and the network forgets its input; with this the logits of different vehicles differ by ~1);
      * embedding tables with per-row scales log-uniform over three decades (row norms 10^-1.5 ... 10^1.5 of the init's);
      * the last layers of the action / RTG heads scaled by `logit_gain`: |logit| reaches 25-35, max probabilities 0.3 ... 0.96.

    Why not more: the settings were tuned on the CPU (float32 against float64 evaluation of the same network).  Random networks
    leave the well-conditioned regime quickly — with gains in [0.3, 3], matrix scales in [0.5, 2] and query / key rows x 3 the
    REFERENCE's own float32 logits differ from the float64 evaluation by 1.4 at |logit| = 43 (two float32 implementations disagree by
    0.5-0.9), so "the reference's logits" stop being a target.  At the settings above float32 is 2-3e-6 of max |logit| away from
    float64: the sharpest regime in which a 1e-5 relative parity bound means something.  No trained checkpoint exists in this
    environment; these are the regimes the fp16-plane operand split (csrc/split.h) has to survive: non-unit gains, dynamic range
    inside a matrix and across embedding rows, peaked softmax, sharp logits."""
    out = generate(dims, seed)
    D = dims.D

    def logu(key, n, lo, hi):
        return np.exp(np.log(lo) + uniform01(key, n) * (np.log(hi) - np.log(lo)))

    for name, shape, kind, _ in param_table(dims):
        key = _fnv1a64(name + "#trained") ^ ((seed * 0x9E3779B97F4A7C15) & _MASK)
        n = int(np.prod(shape))
        v = out[name].astype(np.float64)
        if kind == "ln_w":
            v = logu(key, n, 0.5, 2.0)
        elif kind == "ln_b":
            v = (uniform01(key, n) * 2.0 - 1.0) * 0.15
        elif kind == "bias":
            v = (uniform01(key, n) * 2.0 - 1.0) * 0.1
        elif kind == "normal":                                   # embedding tables: rows over three decades
            v = v * logu(key, shape[0], 10.0 ** -1.5, 10.0 ** 1.5)[:, None]
        elif kind in ("xavier", "uniform") and len(shape) == 2:
            v = v * logu(key, shape[0], 0.7, 1.4)[:, None] * logu(key ^ 0x2545F4914F6CDD1D, shape[1], 0.7, 1.4)[None, :]
            if name.endswith("in_proj_weight"):
                v[:2 * D] *= 2.0
        out[name] = v.astype(np.float32).reshape(shape)
    for head in ("decoder.predict_action.mlp.3", "decoder.predict_rtg.mlp.3"):
        if head + ".weight" in out:
            out[head + ".weight"] = (out[head + ".weight"] * np.float32(logit_gain)).astype(np.float32)
            out[head + ".bias"] = (out[head + ".bias"] * np.float32(logit_gain)).astype(np.float32)
    return out


def from_state_dict(dims: Dims, state_dict) -> dict:
    """Pick and validate the table's tensors from a (Lightning) state_dict of torch tensors/ndarrays."""
    out = {}
    for name, shape, _, _ in param_table(dims):
        if name not in state_dict:
            raise KeyError(f"checkpoint is missing parameter {name}")
        v = state_dict[name]
        v = v.detach().cpu().numpy() if hasattr(v, "detach") else np.asarray(v)
        if tuple(v.shape) != tuple(shape):
            raise ValueError(f"{name}: expected shape {shape}, got {tuple(v.shape)}")
        out[name] = np.ascontiguousarray(v, dtype=np.float32)
    return out


def num_params(dims: Dims) -> int:
    return sum(int(np.prod(s)) for _, s, _, _ in param_table(dims))


def exp_noise(seed: int, scenario: int, t: int, agent: int, head: int, n: int) -> np.ndarray:
    """Exp(1) race noise q for Gumbel-max sampling (token = argmax p/q), keyed by
    (seed, scenario, step, global agent index, head in {0:rtg_goal,1:rtg_veh,2:rtg_road,3:action}).
    float32, q = -log(u) with u the 24-bit uniform above.  The HIP sampler's in-kernel generator
    evaluates the same hash (csrc/sample.hip)."""
    key = noise_key(seed, scenario, t, agent, head)
    return (-np.log(uniform01(key, n))).astype(np.float32)


def noise_key(seed: int, scenario: int, t: int, agent: int, head: int) -> int:
    k = (int(seed) * 0x9E3779B97F4A7C15 + 0x1234567) & _MASK
    for v in (scenario, t, agent, head):
        k = int(splitmix64(np.array([(k ^ (int(v) & _MASK)) & _MASK], dtype=np.uint64))[0])
    return k
