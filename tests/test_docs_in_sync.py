"""DESIGN.md section 5's table and the README's result table are GENERATED from the full bench record committed under profiles/
(tools/doc_numbers.py): no measured figure in them is transcribed by hand.  This test regenerates both and compares."""
import glob
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _between(text, a, b):
    return text[text.index(a) + len(a):text.index(b)].strip()


def test_measured_tables_are_the_generated_ones():
    records = sorted(glob.glob(os.path.join(ROOT, "profiles", "*_bench_details.json")))
    assert len(records) == 1, "profiles/ holds ONE full bench record (one tag): %s" % records
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "doc_numbers.py"), records[0]], capture_output=True, text=True,
                       timeout=120)
    assert r.returncode == 0, r.stderr[-1500:]
    design = _between(open(os.path.join(ROOT, "DESIGN.md")).read(), "<!-- measured:begin -->", "<!-- measured:end -->")
    readme = _between(open(os.path.join(ROOT, "README.md")).read(), "<!-- result:begin -->", "<!-- result:end -->")
    assert design in r.stdout, "DESIGN.md section 5 is not what tools/doc_numbers.py generates from %s" % os.path.basename(records[0])
    assert readme in r.stdout, "README.md's result table is not what tools/doc_numbers.py generates from %s" % os.path.basename(records[0])
    # ... and the committed contract line of the same run is short, strict JSON with the same headline value
    import json
    line = open(records[0].replace("_bench_details.json", "_bench_line.json")).read()
    assert line.count("\n") == 1 and len(line.encode()) <= 6001
    d, full = json.loads(line), json.load(open(records[0]))
    assert d["value"] == full["value"] and d["roofline"]["frac"] == float("%.6g" % full["roofline"]["frac"])
