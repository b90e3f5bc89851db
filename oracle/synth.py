"""Host twin of spx_synth_fill (csrc/spx_ctx.hip): the counter-based synthetic inputs of SURVEY.md 8d, reproduced with numpy
integer arithmetic -- bit-identical to the device generator, no GPU, no torch.  TEST / BENCH INFRASTRUCTURE ONLY."""
import numpy as np

_M64 = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(x):
    with np.errstate(over="ignore"):
        x = x + np.uint64(0x9E3779B97F4A7C15)
        x = (x ^ (x >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        x = (x ^ (x >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        return x ^ (x >> np.uint64(31))


def fill(n, seed, stream, kind, scale=1.0, start=0):
    """out[i] = scale * value(seed, stream, start + i), i < n.  kind 0: U(-1/2, 1/2); kind 1: ~N(0,1) (Irwin-Hall of 12)."""
    with np.errstate(over="ignore"):
        key = _splitmix64(np.uint64(seed) ^ (np.uint64(stream) * np.uint64(0xD1342543DE82EF95)))
        i = np.arange(start, start + n, dtype=np.uint64)
        if kind == 0:
            v = (_splitmix64(key + i) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53 - 0.5
        elif kind == 1:
            acc = np.zeros(n, dtype=np.float64)
            for k in range(12):
                acc += (_splitmix64(key + i * np.uint64(12) + np.uint64(k)) >> np.uint64(11)).astype(np.float64) * 2.0 ** -53
            v = acc - 6.0
        else:
            raise ValueError("kind must be 0 or 1")
    return scale * v
