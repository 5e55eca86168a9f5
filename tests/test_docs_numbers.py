"""The measured numbers quoted in DESIGN.md 5, README.md and profiles/README.md are GENERATED from the tracked files under
profiles/ (tools/gen_docs_numbers.py); this test fails when a document's block and the files disagree (VERDICT r03 item
7: prose that cites numbers the files do not contain)."""
import os
import re
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def test_documents_quote_what_the_tracked_profiles_say():
    import gen_docs_numbers as g
    R = g.latest_round()
    assert R is not None
    assert g.apply(R, check=True) == [], "run: python tools/gen_docs_numbers.py %s" % R
    for doc in g.DOCS:
        text = open(os.path.join(ROOT, doc)).read()
        m = re.search(r"<!-- numbers:(?:measured|summary) (r\d+) -->", text)
        assert m and m.group(1) == R, (doc, "block is not of the newest round with a bench line")


def test_the_block_is_made_of_file_values():
    """spot check: the rocprofv3 average of the headline kernel and the bench line's own numbers appear verbatim"""
    import csv
    import json
    import gen_docs_numbers as g
    R = g.latest_round()
    block = g.measured_block(R)
    b = json.load(open(os.path.join(ROOT, "profiles", "%s_bench.json" % R)))
    assert "%.0f Msamples/s" % b["value"] in block and "**%.3f**" % b["roofline"]["frac"] in block
    rows = [r for r in csv.DictReader(open(os.path.join(ROOT, "profiles", "%s_bench_kernel_stats.csv" % R)))
            if "pfb_kernel_os<256, 1, 14, 4, false>" in r["Name"]]
    assert rows and "%.1f µs over %d launches" % (float(rows[0]["AverageNs"]) * 1e-3, int(rows[0]["Calls"])) in block
