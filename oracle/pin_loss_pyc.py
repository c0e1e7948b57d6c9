"""TEST INFRASTRUCTURE (oracle).  Pins oracle/freqsplit_ref.py (and the ContextualLoss restatement) to the ONLY artefact
of `loss.py` the reference ships: /root/reference/__pycache__/loss.cpython-36.pyc.

CPython 3.10 cannot unmarshal a 3.6 code object (the field order changed), so this file carries a minimal reader of the
3.6 marshal stream -- enough to walk every code object and collect its names, constants and line numbers.  Run in the
BUILD container (needs /root/reference):

    python -m oracle.pin_loss_pyc          # writes tests/golden/loss_pyc_constants.json

The JSON is data extracted from the reference's bytecode (function / class names with their first line, per-function
constants and referenced names); tests/test_oracle_golden.py asserts the restatements against it, without touching
/root/reference.
"""
import json
import os
import struct
import sys

PYC = "/root/reference/__pycache__/loss.cpython-36.pyc"
OUT = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "loss_pyc_constants.json")


class _Reader:
    def __init__(self, data):
        self.d, self.p, self.refs = data, 0, []

    def u8(self):
        self.p += 1
        return self.d[self.p - 1]

    def i32(self):
        self.p += 4
        return struct.unpack_from("<i", self.d, self.p - 4)[0]

    def take(self, n):
        self.p += n
        return self.d[self.p - n:self.p]

    def obj(self):
        code = self.u8()
        flag, t = code & 0x80, chr(code & 0x7F)
        idx = None
        if flag:                       # reserve the slot first: containers may be referenced from inside themselves
            idx = len(self.refs)
            self.refs.append(None)
        if t == "0":
            v = None
        elif t == "N":
            v = None
        elif t == "T":
            v = True
        elif t == "F":
            v = False
        elif t == ".":
            v = Ellipsis
        elif t == "i":
            v = self.i32()
        elif t == "l":
            n = self.i32()
            digits = [struct.unpack_from("<H", self.take(2))[0] for _ in range(abs(n))]
            v = sum(dg << (15 * k) for k, dg in enumerate(digits)) * (1 if n >= 0 else -1)
        elif t == "g":
            v = struct.unpack("<d", self.take(8))[0]
        elif t == "y":
            re_, im = struct.unpack("<dd", self.take(16))
            v = complex(re_, im)
        elif t in "su":
            v = self.take(self.i32())
            v = v.decode("utf-8", "replace") if t == "u" else bytes(v)
        elif t in "tT":
            v = self.take(self.i32()).decode("utf-8", "replace")
        elif t in "aA":
            v = self.take(self.i32()).decode("latin-1")
        elif t in "zZ":
            v = self.take(self.u8()).decode("latin-1")
        elif t == ")":
            v = tuple(self.obj() for _ in range(self.u8()))
        elif t == "(":
            v = tuple(self.obj() for _ in range(self.i32()))
        elif t == "[":
            v = [self.obj() for _ in range(self.i32())]
        elif t in "<>":
            v = frozenset(self.obj() for _ in range(self.i32()))
        elif t == "{":
            v = {}
            while True:
                k = self.obj()
                if k is None and self.d[self.p - 1] == ord("0"):
                    break
                v[k] = self.obj()
        elif t == "r":
            return self.refs[self.i32()]
        elif t == "c":                 # CPython 3.6 code object
            argcount, kwonly, nlocals, stack, flags = (self.i32() for _ in range(5))
            bytecode, consts, names, varnames, freevars, cellvars = (self.obj() for _ in range(6))
            filename, name = self.obj(), self.obj()
            firstlineno = self.i32()
            self.obj()                 # lnotab
            v = dict(kind="code", name=name, filename=filename, firstlineno=firstlineno, argcount=argcount, consts=consts,
                     names=names, varnames=varnames)
        else:
            raise ValueError("marshal type %r at %d" % (t, self.p - 1))
        if idx is not None:
            self.refs[idx] = v
        return v


def _plain(v):
    if isinstance(v, tuple):
        return [_plain(x) for x in v]
    if isinstance(v, str):             # identifiers and short literals only: docstrings are the reference's text, not data
        return v if len(v) <= 48 else "<str:%d chars>" % len(v)
    if isinstance(v, (int, float, bool)) or v is None:
        return v
    return None


def collect(code, prefix=""):
    """-> {qualified name: {line, args, names, consts}} for the module and every nested function / class body."""
    qual = prefix + code["name"]
    out = {qual: {"line": code["firstlineno"], "args": list(code["varnames"][:code["argcount"]]), "names": list(code["names"]),
                  "consts": [_plain(c) for c in code["consts"] if not isinstance(c, dict)]}}
    for c in code["consts"]:
        if isinstance(c, dict) and c.get("kind") == "code":
            out.update(collect(c, "" if code["name"] == "<module>" else qual + "."))
    return out


def main():
    data = open(PYC, "rb").read()
    magic = struct.unpack_from("<H", data, 0)[0]
    assert magic == 3379, magic            # CPython 3.6
    src_mtime, src_size = struct.unpack_from("<II", data, 4)
    top = _Reader(data[12:]).obj()
    table = collect(top)
    table["__header__"] = {"magic": magic, "source_size": src_size, "source": top["filename"]}
    with open(OUT, "w") as f:
        json.dump(table, f, indent=1, sort_keys=True)
    print("wrote", OUT, len(table), "code objects")
    for k in sorted(table):
        if k != "__header__":
            print("%-45s line %-4d consts %s" % (k, table[k]["line"], [c for c in table[k]["consts"] if c is not None][:8]))


if __name__ == "__main__":
    sys.exit(main())
