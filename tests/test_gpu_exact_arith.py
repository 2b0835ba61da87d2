"""The short division / square-root sequences of iris_amd/csrc/iris_exact.h give the IEEE result for EVERY operand of
their domains, and the guarded wrappers of iris_device.h for every operand whatsoever: proved on the GPU by
tools/exact/verify_exact.hip (built by __graft_entry__.build() into iris_amd/lib/verify_exact).

* sqrt, x / PI, 1 / d: all 2^32 bit patterns;
* n / d: all 2^23 x 2^23 = 7.0e13 pairs of significands (inside the domain no intermediate over- or underflows, so powers
  of two factor out exactly), ~30 s on an MI355X, plus 2e9 operands with random exponents, signs, zeros and all-ones
  significands that confirm the exponent argument on the hardware;
* the wrappers (domain test + fallback to the compiler's expansion) on zeros, denormals, infinities, NaNs, negative and
  huge operands, with whole waves inside the domain, one lane outside, and every lane arbitrary.
Together with the bit-exact film tests this is what lets the kernels replace 11- and 16-instruction expansions by 3 + 3
and 8 instructions without changing a bit."""
import os
import re
import subprocess

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "iris_amd", "lib", "verify_exact")


def run(*args, timeout=110):
    assert os.path.exists(BIN), "iris_amd/lib/verify_exact is missing: run __graft_entry__.build()"
    out = subprocess.run([BIN] + list(args), capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-1500:], out.stderr[-1500:])
    return out.stdout


@pytest.mark.parametrize("what,at_least", [("sqrt", 1.8e9), ("divpi", 3.3e9), ("rcp", 1.6e9), ("divexp", 1.5e9), ("wrappers", 1.0e10)])
def test_one_operand_functions_and_wrappers(what, at_least):
    text = run(what)
    m = re.search(r"tested (\d+) operands, (\d+) wrong", text)
    assert m and int(m.group(2)) == 0 and int(m.group(1)) >= at_least, text


def test_division_over_every_pair_of_significands():
    text = run("div", "0", "256", timeout=115)
    lines = [l for l in text.splitlines() if "slice" in l]
    assert len(lines) == 256 and all(l.endswith(" 0 wrong") for l in lines), lines[-3:]
    assert sum(int(re.search(r"tested (\d+)", l).group(1)) for l in lines) == (1 << 46)
