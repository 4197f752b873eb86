"""BooleanHypercube: the row order / rotation of the reference.  TEST INFRASTRUCTURE ONLY.

Follows reference plonkish_backend/src/util/arithmetic/bh.rs:5-153: rows are visited in the order
0, 1, x, x^2, ... of GF(2^k) (an LFSR); `next(b) = (b << 1) ^ ((b >> k) * PRIMITIVE[k])`,
`prev(b) = (b >> 1) ^ ((b & 1) * X_INV[k])`.
"""
PRIMITIVES = [1, 3, 7, 11, 19, 37, 67, 131, 285, 529, 1033, 2053, 4179, 8219, 16427, 32771, 65581, 131081,
              262183, 524327, 1048585, 2097157, 4194307, 8388641, 16777243, 33554441, 67108935, 134217767,
              268435465, 536870917, 1073741907, 2147483657]
X_INVS = [0, 1, 3, 5, 9, 18, 33, 65, 142, 264, 516, 1026, 2089, 4109, 8213, 16385, 32790, 65540, 131091, 262163,
          524292, 1048578, 2097153, 4194320, 8388621, 16777220, 33554467, 67108883, 134217732, 268435458,
          536870953, 1073741828]


class BooleanHypercube:
    def __init__(self, num_vars):
        assert num_vars < 32  # bh.rs:85
        self.num_vars = num_vars
        self.primitive = PRIMITIVES[num_vars]
        self.x_inv = X_INVS[num_vars]

    def next(self, b):
        b <<= 1
        return b ^ ((b >> self.num_vars) * self.primitive)

    def prev(self, b):
        return (b >> 1) ^ ((b & 1) * self.x_inv)

    def rotate(self, b, rotation):
        """bh.rs:108-125"""
        for _ in range(-rotation if rotation < 0 else 0):
            b = self.prev(b)
        for _ in range(rotation if rotation > 0 else 0):
            b = self.next(b)
        return b

    def iter(self):
        """bh.rs:127-133: 0, then 1 and its successors, 2^k entries in total"""
        out, b = [0], 1
        while len(out) < 1 << self.num_vars:
            out.append(b)
            b = self.next(b)
        return out

    def nth_map(self):
        out = [0] * (1 << self.num_vars)
        for nth, b in enumerate(self.iter()):
            out[b] = nth
        return out

    def rotation_map(self, rotation):
        return [self.rotate(b, rotation) for b in range(1 << self.num_vars)]
