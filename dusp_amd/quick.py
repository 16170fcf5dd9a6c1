"""Operator helpers (reference src/quick.js:15-110): fold when both operands are
numbers, otherwise build the unit.  Only the helpers whose units the GPU path
executes build anything; the rest fold numbers and refuse signals."""
import numbers

from .graph import Multiply, Sum


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


def _numbers_only(name, fn):
    def helper(*args):
        if all(_num(a) for a in args):
            return fn(*args)
        raise NotImplementedError("quick.%s on signals needs a unit the GPU path does not execute yet" % name)
    helper.__name__ = name
    return helper


subtract = _numbers_only("subtract", lambda a, b: a - b)
divide = _numbers_only("divide", lambda a, b: a / b)
invert = _numbers_only("invert", lambda a: -a)
semitoneToRatio = _numbers_only("semitoneToRatio", lambda p: 2 ** (p / 12))
pToF = _numbers_only("pToF", lambda p: 2 ** ((p - 69) / 12) * 440)
pow = _numbers_only("pow", lambda a, b: a ** b)
clipAbove = _numbers_only("clipAbove", lambda x, th: th if x > th else x)
clipBelow = _numbers_only("clipBelow", lambda x, th: th if x < th else x)
