"""the lines the front-end prints, against the reference's format strings -- no GPU needed.

tools/cli_report.h holds one function per report of the reference's main.cpp (:469 scene, :512-515 grid, :523-533 memory,
:434-444 benchmark).  (1) A small program prints them for fixed numbers and the text must equal what the reference's statements
print for those numbers (written out below from main.cpp's format strings; floating-point numbers go through operator<< with the
default precision on both sides).  (2) Where the reference checkout exists, the string literals of those statements are taken
from main.cpp itself and must appear, in order, in the output; the usage text must start with the reference's usage text."""
import os
import re
import subprocess
import tempfile

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REF_MAIN = "/root/reference/src/main.cpp"

PROG = r'''
#include <iostream>
#include "cli_report.h"
int main() {
    hagrid_cli::report_scene(std::cout, 1000000);
    hagrid_cli::report_grid(std::cout, 4.4412, 200, 200, 200, 4227724, 5580311);
    hagrid_cli::report_memory(std::cout, size_t(4227724) * 32, size_t(7659232) * 4, size_t(5580311) * 4, size_t(1000000) * 48, 912345678);
    hagrid_cli::report_timings(std::cout, {0.25, 0.125, 0.5}, 1048576, 853017);
    hagrid_cli::report_grid(std::cout, -1.0, 32, 32, 32, 16398, 24753);
    return 0;
}
'''

# what main.cpp:469, :512-515, :523-533 and :434-444 print for the numbers above (default ostream formatting: 6 significant digits)
EXPECTED = """1000000 triangle(s)
Grid built in 4.4412 ms (200x200x200, 4227724 cells, 5580311 references)
Total memory: 225.301 MB
Cells: 129.02 MB
Entries: 29.2177 MB
References: 21.2872 MB
Triangles: 45.7764 MB
Peak usage: 870.081 MB
853017 intersection(s).
0.875ms for 3 iteration(s).
3595.12 Mrays/sec.
# Average: 0.291667 ms
# Median: 0.25 ms
# Min: 0.125 ms
Grid loaded (32x32x32, 16398 cells, 24753 references)
"""


@pytest.fixture(scope="module")
def report_text():
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.cpp"), "w").write(PROG)
        exe = os.path.join(d, "p")
        subprocess.run(["g++", "-std=c++11", "-O1", "-Wall", "-I", os.path.join(ROOT, "tools"), os.path.join(d, "p.cpp"), "-o", exe], check=True)
        return subprocess.run([exe], capture_output=True, text=True, check=True).stdout


def test_report_lines_for_fixed_numbers(report_text):
    assert report_text == EXPECTED


def _cout_literals(text):
    """The string literals of the std::cout statements in `text`, in order."""
    out = []
    for stmt in re.findall(r"std::cout\s*<<(.*?);", text, flags=re.S):
        out += re.findall(r'"((?:[^"\\]|\\.)*)"', stmt)
    return [s for s in out if s.strip()]


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference checkout not present")
def test_literals_of_the_reference_statements_appear_in_order(report_text):
    lines = open(REF_MAIN).read().split("\n")
    pick = lambda a, b: "\n".join(lines[a - 1:b])
    # (the order of the calls in PROG: scene, grid, memory, timings)
    lits = _cout_literals(pick(469, 469)) + _cout_literals(pick(512, 515)) + _cout_literals(pick(523, 533)) + _cout_literals(pick(439, 444))
    assert len(lits) >= 25, lits
    at = 0
    for lit in lits:
        lit = lit.encode().decode("unicode_escape")
        found = report_text.find(lit, at)
        assert found >= 0, (lit, report_text[at:at + 200])
        at = found + len(lit)


@pytest.mark.skipif(not os.path.exists(REF_MAIN), reason="reference checkout not present")
def test_usage_text_starts_with_the_reference_usage():
    import test_cpp_api as T
    src = open(REF_MAIN).read()
    body = src[src.index("static void usage()"):]
    body = body[:body.index("<< std::endl")]
    want = "".join(s.encode().decode("unicode_escape") for s in re.findall(r'"((?:[^"\\]|\\.)*)"', body))
    assert want.startswith("Usage: hagrid [options] file\n") and want.count("\n") >= 20
    with tempfile.TemporaryDirectory() as d:
        exe = T._build_cli(d)
        got = subprocess.run([exe, "--help"], capture_output=True, text=True, check=True).stdout
    assert got.startswith(want), (got, want)
    assert "Extensions of this front-end" in got[len(want):]


def test_usage_sections_without_the_reference():
    import test_cpp_api as T
    with tempfile.TemporaryDirectory() as d:
        exe = T._build_cli(d)
        got = subprocess.run([exe, "--help"], capture_output=True, text=True, check=True).stdout.split("\n")
    assert got[0] == "Usage: hagrid [options] file" and got[1] == "Options:" and got[2] == "  -h      --help          Shows this message"
    assert " Construction parameters:" in got and " Benchmarking:" in got
    assert got.index(" Construction parameters:") < got.index("  -td     --top-density   Sets the top-level density") < got.index(" Benchmarking:")
