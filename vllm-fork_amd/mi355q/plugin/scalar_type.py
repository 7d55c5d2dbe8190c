"""Sub-byte weight types as the reference names them (vllm/scalar_type.py:22-…, mirrored in
csrc/core/scalar_type.hpp): an integer of `size_bits` bits stored as `value + bias`.
Only what the W4A16/W8A16 path needs: uint4 (AWQ, explicit zero points), uint4b8 and
uint8b128 (GPTQ symmetric, zero point folded into the bias)."""
from dataclasses import dataclass


@dataclass(frozen=True)
class ScalarType:
    exponent: int
    mantissa: int
    signed: bool
    bias: int = 0

    @property
    def size_bits(self) -> int:
        return self.exponent + self.mantissa + int(self.signed)

    def is_integer(self) -> bool:
        return self.exponent == 0

    def has_bias(self) -> bool:
        return self.bias != 0

    def min(self) -> int:
        if not self.is_integer():
            raise NotImplementedError("only integer types are modelled")
        return (-(1 << self.mantissa) if self.signed else 0) - self.bias

    def max(self) -> int:
        if not self.is_integer():
            raise NotImplementedError("only integer types are modelled")
        return (1 << self.mantissa) - 1 - self.bias

    @property
    def id(self) -> int:
        """Stable integer handle handed across op boundaries (the reference packs the same
        four fields into an int64, scalar_type.py `id`)."""
        return self.exponent | (self.mantissa << 8) | (int(self.signed) << 16) | ((self.bias & 0xFFFFFFFF) << 17)

    def __str__(self) -> str:
        s = ("int" if self.signed else "uint") + str(self.size_bits)
        return s + (f"b{self.bias}" if self.bias else "")

    __repr__ = __str__

    @classmethod
    def uint(cls, size_bits: int, bias: int | None = None) -> "ScalarType":
        return cls(0, size_bits, False, bias or 0)


class scalar_types:
    uint4 = ScalarType.uint(4)
    uint8 = ScalarType.uint(8)
    uint4b8 = ScalarType.uint(4, 8)
    uint8b128 = ScalarType.uint(8, 128)
