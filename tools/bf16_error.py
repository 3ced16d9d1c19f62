"""How far reduced-precision operands (f32 accumulation) move the network's logits from an f64
evaluation, next to plain f32, for random weights ~U(-0.5, 0.5) (lib.rs:769-781).  CPU only,
numpy.  Variants:
  bf16        operands rounded to bf16, one product per term
  bf16x2/3p   operands split a = hi + lo (two bf16), products hi*hi + hi*lo + lo*hi
  bf16x3/6p   operands split into three bf16 pieces (24 significand bits, i.e. the whole f32
              value), the six products whose weight is >= 2^-16 of the leading one
  bf16x3/9p   all nine products
Every partial product of two bf16 values is exact in f32; the sums are f32 like the MFMA
accumulator's."""
import numpy as np


def bf16(x):
    u = x.astype(np.float32).view(np.uint32)
    return ((u + ((u >> 16) & 1) + 0x7FFF) & 0xFFFF0000).view(np.float32)


def pieces(x, n):
    out, r = [], x.astype(np.float32)
    for _ in range(n):
        p = bf16(r)
        out.append(p)
        r = (r - p).astype(np.float32)
    return out


def split_matmul(n, keep):
    """keep(i, j) -> whether piece i of the left operand times piece j of the right is computed"""
    def mm(a, b):
        pa, pb = pieces(a, n), pieces(b, n)
        acc = np.zeros((a.shape[0], b.shape[1]), np.float32)
        # smallest terms first, as a kernel that wants the tightest f32 sum would order them
        for i, j in sorted(((i, j) for i in range(n) for j in range(n) if keep(i, j)), key=lambda t: -(t[0] + t[1])):
            acc = (acc + pa[i] @ pb[j]).astype(np.float32)
        return acc
    return mm


VARIANTS = {
    "f32": lambda a, b: a @ b,
    "bf16": lambda a, b: bf16(a) @ bf16(b),
    "bf16x2/3p": split_matmul(2, lambda i, j: i + j <= 1),
    "bf16x3/6p": split_matmul(3, lambda i, j: i + j <= 2),
    "bf16x3/9p": split_matmul(3, lambda i, j: True),
}


def main():
    rng = np.random.default_rng(0)
    for dims in ((60, 512, 256, 16), (60, 2048, 2048, 256)):
        i, h1, h2, c = dims
        w1, w2, w3 = (rng.uniform(-0.5, 0.5, s).astype(np.float32) for s in ((i, h1), (h1, h2), (h2, c)))
        x = rng.standard_normal((256, i)).astype(np.float32)

        def logits(mm, dt):
            a = np.maximum(mm(x.astype(dt), w1.astype(dt)), 0).astype(dt)
            b = np.tanh(mm(a, w2.astype(dt))).astype(dt)
            return mm(b, w3.astype(dt)).astype(np.float64), a.astype(np.float64), b.astype(np.float64)

        ref = logits(lambda a, b: a @ b, np.float64)
        print(dims, "max |error| against f64   (logits, hidden 1 pre-activation scale %.1f, tanh layer)" % np.abs(ref[1]).max())
        for name, mm in VARIANTS.items():
            got = logits(mm, np.float32)
            print("  %-10s logits %.2e   h1 %.2e   h2 %.2e" % ((name,) + tuple(np.abs(g - r).max() for g, r in zip(got, ref))))


if __name__ == "__main__":
    main()
