"""Every behaviour switch of the context (FH_OPTION_LIST, fidget_amd/csrc/capi_core.hpp) is named in DESIGN.md section 5's list: an option
nobody can look up is an option nobody can use."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_context_option_is_in_the_design_document():
    src = open(os.path.join(ROOT, "fidget_amd", "csrc", "capi_core.hpp")).read()
    m = re.search(r"#define FH_OPTION_LIST\(X\)(.*?)\nstruct FhOptions", src, re.S)
    names = re.findall(r"X\((\w+),", m.group(1))
    assert len(names) > 40
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [n for n in names if f"`{n}`" not in doc]
    assert not missing, missing
