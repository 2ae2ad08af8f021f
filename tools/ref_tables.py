#!/usr/bin/env python3
"""Reads the H.266 constant tables straight out of the reference's libavcodec/vvc/vvc_data.c as data (a small C-initialiser reader
written for this purpose: brace initialisers, the FILTER_G() rows and the DEFINE_DCT8 / DEFINE_DST7 matrix macros expanded), in the
flat layout of this repo's exported tables.  Independent of tools/gen_tables.py and of ffvvc_amd/csrc/tables.inc: the table check
(tests/test_tables_cpu.py) compares what the built library exports against these values, and against SHA-256 digests of them that
are committed as a fixture (tests/golden/tables_sha256.json; this script writes it when run in the build container)."""
import hashlib
import json
import os
import re
import struct

REF = "/root/reference/libavcodec/vvc/vvc_data.c"
FIXTURE = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "tables_sha256.json")


def _strip(text):
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return re.sub(r"//[^\n]*", "", text)


def _initialiser(text, name):
    """Text between the outermost braces of `name`'s initialiser."""
    m = re.search(r"\b%s\b[^;{=]*=\s*\{" % re.escape(name), text)
    if m is None:
        raise KeyError(name)
    depth, i = 1, m.end()
    while depth:
        depth += {"{": 1, "}": -1}.get(text[i], 0)
        i += 1
    return text[m.end():i - 1]


def _numbers(body):
    return [int(t) for t in re.findall(r"(?<![\w.])-?\d+\b", body)]


def _macro(text, name):
    """(parameter names, body) of a function-like macro with line continuations."""
    m = re.search(r"#define\s+%s\(([^)]*)\)((?:[^\n]*\\\n)*[^\n]*)" % re.escape(name), text)
    params = [p.strip() for p in m.group(1).split(",")]
    return params, m.group(2).replace("\\\n", " ")


def _expand_matrix(text, table):
    """An int8 matrix defined as `table = DEFINE_xxx(values...)`: substitute the values for the macro's letters."""
    m = re.search(r"\b%s\b[^=;]*=\s*(\w+)\s*\(([^)]*)\)" % re.escape(table), text)
    params, body = _macro(text, m.group(1))
    vals = dict(zip(params, [int(v) for v in m.group(2).split(",")]))
    out = []
    for tok in re.findall(r"-?\s*\b[A-Za-z]\b|-?\s*\b\d+\b", body):
        tok = tok.replace(" ", "")
        neg = tok.startswith("-")
        key = tok.lstrip("-")
        v = vals[key] if key in vals else int(key)
        out.append(-v if neg else v)
    return out


def read_reference(path=REF):
    text = _strip(open(path).read())
    t = {}
    for n in (4, 8, 16, 32):
        t[f"dst7_{n}"] = ("b", _expand_matrix(text, f"ff_vvc_dst7_{n}x{n}"))
        t[f"dct8_{n}"] = ("b", _expand_matrix(text, f"ff_vvc_dct8_{n}x{n}"))
    t["inter_luma_filters"] = ("b", _numbers(_initialiser(text, "ff_vvc_inter_luma_filters")))
    t["inter_chroma_filters"] = ("b", _numbers(_initialiser(text, "ff_vvc_inter_chroma_filters")))
    body = _initialiser(text, "ff_vvc_intra_luma_filter")
    fc = _numbers(re.sub(r"FILTER_G\s*\(\s*\d+\s*\)", "", body))
    fg = [v for p in (int(x) for x in re.findall(r"FILTER_G\s*\(\s*(\d+)\s*\)", body)) for v in (16 - (p >> 1), 32 - (p >> 1), 16 + (p >> 1), p >> 1)]
    t["intra_luma_filter"] = ("b", fc + fg)
    t["lfnst_8x8"] = ("b", _numbers(_initialiser(text, "ff_vvc_lfnst_8x8")))
    t["lfnst_4x4"] = ("b", _numbers(_initialiser(text, "ff_vvc_lfnst_4x4")))
    t["lfnst_tr_set_index"] = ("B", _numbers(_initialiser(text, "ff_vvc_lfnst_tr_set_index")))
    t["alf_fix_filt_coeff"] = ("h", _numbers(_initialiser(text, "ff_vvc_alf_fix_filt_coeff")))
    t["alf_class_to_filt_map"] = ("B", _numbers(_initialiser(text, "ff_vvc_alf_class_to_filt_map")))
    t["alf_aps_class_to_filt_map"] = ("B", _numbers(_initialiser(text, "ff_vvc_alf_aps_class_to_filt_map")))
    for n in ("mip_matrix_4x4", "mip_matrix_8x8", "mip_matrix_16x16"):
        t[n] = ("B", _numbers(_initialiser(text, n)))
    # ---- the tables the reference keeps inline in other files (function-local `static const` arrays)
    d = os.path.dirname(path)
    flt, intra, intra_t, flt_t = (_strip(open(os.path.join(d, f)).read()) for f in ("vvc_filter.c", "vvc_intra.c", "vvc_intra_template.c", "vvc_filter_template.c"))
    t["tc_table"] = ("H", _numbers(_initialiser(flt, "tctable")))                              # vvc_filter.c:38
    t["beta_table"] = ("B", _numbers(_initialiser(flt, "betatable")))                          # :47
    t["intra_angles"] = ("h", _numbers(_initialiser(intra, "angles")))                         # vvc_intra.c:667
    t["level_scale"] = ("B", _numbers(_initialiser(intra, "level_scale")))                     # :329
    t["ref_filter_modes"] = ("b", _numbers(_initialiser(intra, "modes").replace("INTRA_PLANAR", "0")))      # :657 (INTRA_PLANAR = 0, vvc_ctu.h)
    t["intra_filter_thres"] = ("B", _numbers(_initialiser(intra_t, "intra_hor_ver_dist_thres")))            # vvc_intra_template.c:559
    t["cclm_div_sig"] = ("B", _numbers(_initialiser(intra_t, "div_sig_table")))                # :261
    t["alf_arg_var"] = ("B", _numbers(_initialiser(flt_t, "arg_var")))                         # vvc_filter_template.c:272
    t["alf_transpose_index"] = ("B", _numbers(_initialiser(flt_t, "index")))                   # :387
    for axis in "xy":
        t[f"diag_scan_4x4_{axis}"] = ("B", _sub_initialiser(_initialiser(text, f"ff_vvc_diag_scan_{axis}"), (2, 2)))   # vvc_data.c:27,152: [log2 w][log2 h]
    return t


def _sub_initialiser(body, index):
    """Numbers of the sub-aggregate body[index[0]][index[1]]... of a nested brace initialiser (rows may be shorter than their bound)."""
    for want in index:
        depth, seen, start = 0, -1, None
        for i, ch in enumerate(body):
            if ch == "{":
                if depth == 0:
                    seen += 1
                    if seen == want:
                        start = i + 1
                depth += 1
            elif ch == "}":
                depth -= 1
                if depth == 0 and start is not None:
                    body = body[start:i]
                    break
        else:
            raise IndexError(index)
    return _numbers(body)


def digest(fmt, values):
    return hashlib.sha256(struct.pack("<%d%s" % (len(values), fmt), *values)).hexdigest()


def main():
    t = read_reference()
    fix = {name: {"type": {"b": "int8", "B": "uint8", "h": "int16", "H": "uint16"}[fmt], "count": len(v), "sha256": digest(fmt, v)} for name, (fmt, v) in sorted(t.items())}
    with open(FIXTURE, "w") as f:
        json.dump({"source": "libavcodec/vvc/vvc_data.c, vvc_filter.c, vvc_intra.c, vvc_intra_template.c, vvc_filter_template.c of the reference, read by tools/ref_tables.py", "tables": fix}, f, indent=1)
        f.write("\n")
    print("wrote", FIXTURE, len(fix), "tables")


if __name__ == "__main__":
    main()
