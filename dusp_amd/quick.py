"""Operator helpers (reference src/quick.js:15-110): fold when both operands are
numbers, otherwise build the unit."""
import numbers

from .graph import ConcatChannels, Divide, HardClipAbove, HardClipBelow, Multiply, PolarityInvert, Pow, SemitoneToRatio, Subtract, Sum


def _num(x):
    return isinstance(x, numbers.Number)


def add(a, b):
    return a + b if _num(a) and _num(b) else Sum(a, b)


def mult(a, b):
    if a is None or (_num(a) and a == 1):
        return b
    if b is None or (_num(b) and b == 1):
        return a
    return a * b if _num(a) and _num(b) else Multiply(a, b)


multiply = mult


def _is_signal(x):
    return getattr(x, "isUnit", False) or getattr(x, "isOutlet", False)


def subtract(a, b):
    return a - b if _num(a) and _num(b) else Subtract(a, b)


def divide(a, b):
    return a / b if _num(a) and _num(b) else Divide(a, b)


def invert(a):
    return -a if _num(a) else PolarityInvert(a)


def semitoneToRatio(p):
    return 2 ** (p / 12) if _num(p) else SemitoneToRatio(p)


def pToF(p):
    if _num(p):
        return 2 ** ((p - 69) / 12) * 440
    raise NotImplementedError("quick.pToF(non number) has not been implemented")  # the reference's message (quick.js:55)


def pow(a, b):  # noqa: A001 - the reference's name
    return Pow(a, b) if _is_signal(a) or _is_signal(b) else a ** b


def clipAbove(input, threshold):
    if _is_signal(input) or _is_signal(threshold):
        return HardClipAbove(input, threshold)
    return threshold if input > threshold else input


def clipBelow(input, threshold):
    if _is_signal(input) or _is_signal(threshold):
        return HardClipBelow(input, threshold)
    return threshold if input < threshold else input


def concat(a, b):  # quick.js:68-73
    if _is_signal(a) or _is_signal(b):
        return ConcatChannels(a, b)
    as_list = lambda x: list(x) if isinstance(x, (list, tuple)) else [x]
    return as_list(a) + as_list(b)
