"""TEST INFRASTRUCTURE.  Prints the 64-entry table of pm_log (oracle/mcgpu_oracle.c and the COMPAT kernel's track_common.inc):
the argument's bits minus OFF = bits(0x3FE6A09E00000000 ~ sqrt(1/2)) give k = exponent, i = the top 6 bits below it and
z = x 2^-k in [OFF, 2 OFF); interval i has the centre c_i (exactly 1 for the interval that contains 1.0, so that arguments
near 1 lose nothing to cancellation) and the table holds {invc = double(1 / c_i), logc = double(-ln(invc))}: with
r = fma(z, invc, -1), ln x = k ln2 + logc + log1p(r) holds exactly in real arithmetic for the ROUNDED invc, |r| < 0.0157.
Pure arithmetic (decimal, 60 digits): nothing is read from any library."""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60
OFF = 0x3FE6A09E00000000
N_BITS = 6


def as_double(bits):
    return struct.unpack("<d", struct.pack("<Q", bits))[0]


def table():
    out = []
    step = 1 << (52 - N_BITS)
    for i in range(1 << N_BITS):
        lo, hi = as_double(OFF + i * step), as_double(OFF + (i + 1) * step)
        c = Decimal(1) if lo <= 1.0 < hi else (Decimal(lo) + Decimal(hi)) / 2
        invc = float(Decimal(1) / c)
        logc = float(-(Decimal(invc).ln()))
        out.append((invc, 0.0 if logc == 0 else logc))
    return out


if __name__ == "__main__":
    t = table()
    for j in range(0, len(t), 2):
        print("  " + ", ".join("{%s, %s}" % (a.hex(), b.hex()) for a, b in t[j:j + 2]) + ",")
    worst = 0.0
    step = 1 << (52 - N_BITS)
    for i, (invc, _) in enumerate(t):
        for z in (as_double(OFF + i * step), as_double(OFF + (i + 1) * step - 1)):
            worst = max(worst, abs(z * invc - 1.0))
    print("/* max |r| = %.6f */" % worst)
