"""Documentation integrity: every evidence file, tool and test the design documents name exists in the tree."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
DOCS = ["DESIGN.md", "README.md", "INTEGRATION.md", "profiles/README.md", "docs/design_history_r3.md", "tools/README.md"]


def _text():
    return {d: open(os.path.join(ROOT, d)).read() for d in DOCS}


def test_every_profile_file_named_in_the_documents_exists():
    have = {os.path.basename(f) for f in glob.glob(os.path.join(ROOT, "profiles", "*"))}
    missing = []
    for doc, text in _text().items():
        for name in set(re.findall(r"\b(r0\d_[A-Za-z0-9_]+\.(?:csv|json|txt))\b", text)):
            if name not in have:
                missing.append((doc, name))
    assert not missing, missing


def test_every_tool_and_test_named_in_the_documents_exists():
    missing = []
    test_src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "tests", "*.py")) + glob.glob(os.path.join(ROOT, "tests", "*", "*.py")))
    for doc, text in _text().items():
        for path in set(re.findall(r"`((?:tools|tests|examples|julia|include|oracle|docs)/[A-Za-z0-9_./-]+\.(?:py|sh|hip|json|md|c|jl|h))", text)):
            if not os.path.exists(os.path.join(ROOT, path)):
                missing.append((doc, path))
        for name in set(re.findall(r"`(test_[a-z0-9_]+)`", text)):
            if ("def %s(" % name) not in test_src:
                missing.append((doc, name))
    assert not missing, missing


def test_every_kernel_and_entry_point_named_in_the_documents_exists():
    """`…_kernel` names in the design documents are kernels of `benlsip.jl_amd/csrc`, `bh_…` names are exports of the header (or
    options / struct members documented there)."""
    src = "".join(open(f).read() for f in glob.glob(os.path.join(ROOT, "benlsip.jl_amd", "csrc", "*")))
    header = open(os.path.join(ROOT, "include", "benlsip_hip.h")).read()
    missing = []
    for doc, text in _text().items():
        if doc.startswith("docs/design_history"):
            continue                                    # the engineering log also names kernels that were tried and removed
        for name in set(re.findall(r"`([a-z][a-z0-9_]*_kernel)\b", text)):
            if not re.search(r"\b%s\b" % re.escape(name), src):
                missing.append((doc, name))
        for name in set(re.findall(r"`(bh_[a-z0-9_]+)`", text)):
            if not re.search(r"\b%s\b" % re.escape(name), header):
                missing.append((doc, name))
    assert not missing, missing
