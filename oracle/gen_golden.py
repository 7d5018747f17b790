#!/usr/bin/env python3
"""TEST INFRASTRUCTURE: generate tests/golden/codegen_vectors.json from the REFERENCE's own
closed-form CasADi codegen (/root/reference/src/Codegen/rev_*_{FD,ID}.cpp, compiled by
`make -C oracle ref` into oracle/_ref/).  Runs only where /root/reference exists; the JSON it
writes (inputs + expected outputs, hex floats) is the committed fixture.

Models: uniform RevoluteChainWithRotor<2,4> and RevolutePairChainWithRotor<2,4>;
the reference checks its cluster ABA / RNEA against exactly these functions in
UnitTests/testReflectedInertiaAlgos.cpp:144-222.
"""
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import oracle_py as O  # noqa: E402

N_STATES = 64


def main():
    if not O.ref_available():
        raise SystemExit("oracle/_ref/libgrbda_codegen_ref.so missing: run `make -C oracle ref` first")
    rng = np.random.Generator(np.random.Philox(0x67726264))
    out = {"source": "reference src/Codegen via oracle/_ref (CasADi 3.6.3 generated closed forms)", "cases": []}
    for fam in ("rev", "pair"):
        for n in (2, 4):
            y, yd, x = (rng.uniform(-1.0, 1.0, size=(N_STATES, n)) for _ in range(3))
            fd = O.ref_codegen(fam, n, "FD", y, yd, x)
            idd = O.ref_codegen(fam, n, "ID", y, yd, x)
            hexa = lambda a: [[float(v).hex() for v in row] for row in a]
            out["cases"].append({"family": fam, "n": n, "y": hexa(y), "yd": hexa(yd), "x": hexa(x),
                                 "FD": hexa(fd), "ID": hexa(idd)})
    path = os.path.join(HERE, "..", "tests", "golden", "codegen_vectors.json")
    with open(path, "w") as f:
        json.dump(out, f)
    print("wrote", os.path.normpath(path))


if __name__ == "__main__":
    main()
