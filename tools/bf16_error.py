"""How far bf16 operands (f32 accumulation) move the network's logits from an f64 evaluation,
next to plain f32, for random weights ~U(-0.5, 0.5) (lib.rs:769-781).  CPU only, numpy."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    return ((u + ((u >> 16) & 1) + 0x7FFF) & 0xFFFF0000).view(np.float32)


def main():
    rng = np.random.default_rng(0)
    for dims in ((60, 512, 256, 16), (60, 2048, 2048, 256)):
        i, h1, h2, c = dims
        w1, w2, w3 = (rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in ((i, h1), (h1, h2), (h2, c)))
        x = rng.standard_normal((256, i)).astype(np.float32)

        def logits(q):
            a = np.maximum(q(x) @ q(w1), 0).astype(np.float32)
            b = np.tanh(q(a) @ q(w2)).astype(np.float32)
            return (q(b) @ q(w3)).astype(np.float64)

        ref = logits(lambda v: v.astype(np.float64))
        print(dims, "max |logit error|: f32 %.2e, bf16 operands %.2e" %
              (np.abs(logits(lambda v: v) - ref).max(), np.abs(logits(bf16) - ref).max()))


if __name__ == "__main__":
    main()
