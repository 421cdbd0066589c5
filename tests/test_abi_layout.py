"""include/mpcqp.h is the contract; pympc_amd/_lib.py and the ctypes snippet in INTEGRATION.md are hand-written mirrors of
its structs.  A field added to the header and forgotten in a mirror makes mpcqp_default_settings write past the Python
object (that happened: soft_constraints) -- so the struct layouts are parsed out of the header and compared field by field,
and every function prototype of the header must be bound by _lib.SYMBOLS and vice versa."""
import ctypes as C
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = open(os.path.join(ROOT, 'include', 'mpcqp.h')).read()


def _strip_comments(text):
    return re.sub(r'/\*.*?\*/', '', text, flags=re.S)


def header_struct(name):
    """[(field, ctype-kind)] of `typedef struct { ... } name;` with kind in {'double', 'int32', 'ptr'}."""
    body = re.search(r'typedef struct \{([^{}]*)\}\s*%s\s*;' % name, _strip_comments(HEADER)).group(1)
    fields = []
    for decl in body.split(';'):
        decl = ' '.join(decl.split())
        if not decl:
            continue
        m = re.match(r'(const )?(double|int32_t)\s*(.*)', decl)
        assert m, decl
        base = m.group(2)
        for item in m.group(3).split(','):
            item = item.strip()
            ptr = item.startswith('*')
            fields.append((item.lstrip('* ').strip(), 'ptr' if ptr else ('double' if base == 'double' else 'int32')))
    return fields


def ctypes_struct(cls):
    kind = lambda t: 'double' if t is C.c_double else ('int32' if t in (C.c_int32, C.c_int) else 'ptr')
    return [(n, kind(t)) for n, t in cls._fields_]


def header_functions():
    text = _strip_comments(HEADER)
    text = re.sub(r'typedef struct \{.*?\}\s*\w+\s*;', '', text, flags=re.S)
    return sorted(set(re.findall(r'\b(mpcqp_\w+)\s*\(', text)))


def test_ctypes_structs_mirror_the_header():
    from pympc_amd import _lib
    assert ctypes_struct(_lib.Settings) == header_struct('mpcqp_settings')
    assert ctypes_struct(_lib.Info) == header_struct('mpcqp_info')
    assert ctypes_struct(_lib.Model) == header_struct('mpcqp_model')
    assert ctypes_struct(_lib.Loop) == header_struct('mpcqp_loop')
    # sizes as the C compiler lays them out (natural alignment; no packing pragmas in the header)
    assert C.sizeof(_lib.Settings) == 8 * 8 + 9 * 4 + 4          # 9 int32 + tail padding to 8
    assert C.sizeof(_lib.Info) == 4 * 4 + 4 * 8


def test_symbol_list_is_the_header():
    from pympc_amd import _lib
    assert sorted(_lib.SYMBOLS) == header_functions()


def test_integration_doc_snippet_mirrors_the_header():
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    snip = doc[doc.index('class Settings(C.Structure)'):doc.index('s = Settings()')]
    ns = {'C': C}
    exec(snip, ns)
    assert ctypes_struct(ns['Settings']) == header_struct('mpcqp_settings')
    assert ctypes_struct(ns['Model']) == header_struct('mpcqp_model')
