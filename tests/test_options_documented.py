"""Every behaviour switch of the context (FH_OPTION_LIST, fidget_amd/csrc/capi_core.hpp) is named in DESIGN.md section 5's list: an option
nobody can look up is an option nobody can use.  And: no more than 25 of them - an experiment that lost is deleted, not
left behind a switch - and DESIGN.md is the CURRENT design in at most 40 KB, with the rounds' history in a file of its own."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _options():
    src = open(os.path.join(ROOT, "fidget_amd", "csrc", "capi_core.hpp")).read()
    m = re.search(r"#define FH_OPTION_LIST\(X\)(.*?)\nstruct FhOptions", src, re.S)
    return re.findall(r"X\((\w+),", m.group(1))


def test_every_context_option_is_in_the_design_document():
    names = _options()
    assert 15 <= len(names) <= 25, len(names)
    doc = open(os.path.join(ROOT, "DESIGN.md")).read()
    missing = [n for n in names if f"`{n}`" not in doc]
    assert not missing, missing


def test_no_option_is_read_that_the_list_does_not_have():
    names = set(_options())
    used = set()
    for f in os.listdir(os.path.join(ROOT, "fidget_amd", "csrc")):
        if f.endswith((".hpp", ".hip")):
            used |= set(re.findall(r"opt\.(\w+)\b", open(os.path.join(ROOT, "fidget_amd", "csrc", f)).read()))
    used -= {"h", "hpp"}
    assert used <= names, used - names
    assert names <= used | {"probe", "stats"}, names - used          # (every switch is read somewhere)


def test_the_design_document_is_the_current_design():
    assert os.path.getsize(os.path.join(ROOT, "DESIGN.md")) <= 40 * 1024
    assert os.path.exists(os.path.join(ROOT, "DESIGN_HISTORY.md"))
