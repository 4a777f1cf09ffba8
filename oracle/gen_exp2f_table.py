"""TEST INFRASTRUCTURE.  Prints the 32-entry table of the single-precision exp algorithm restated in oracle/mcgpu_oracle.c
(gl_expf) and in the COMPAT kernel (track_common.inc): T[i] = bits(2^(i/32) correctly rounded to double) - (i << 47), so that
adding (k << 47) to T[k % 32] yields the bits of 2^(k/32).  Pure arithmetic: nothing is read from any library."""
import struct
from decimal import Decimal, getcontext

getcontext().prec = 60


def table():
    out = []
    for i in range(32):
        v = float(Decimal(2) ** (Decimal(i) / 32))  # Decimal -> float conversion rounds correctly
        out.append(struct.unpack("<Q", struct.pack("<d", v))[0] - (i << 47))
    return out


if __name__ == "__main__":
    t = table()
    for j in range(0, 32, 6):
        print("  " + ", ".join(f"0x{v:016x}" for v in t[j:j + 6]) + ",")
