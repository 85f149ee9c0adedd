"""The reference-side binding INTEGRATION.md shows (the Zig `extern fn` block a maintainer would add next to
src/MultiIndex.zig:287-330) has never met a Zig compiler -- there is none in this image.  What can be checked without one:
every `pub extern fn` is declared once, names a function include/fpx.h declares, and takes as many parameters; every opaque
handle type is declared once; and the header itself is plain C -- it compiles as C99 and a C caller links against libfpx.so."""
import os
import re
import shutil
import subprocess

import pytest

from fpx_testlib import ROOT

HEADER = os.path.join(ROOT, "include", "fpx.h")


def _split_top_level(params):
    out, depth, cur = [], 0, ""
    for ch in params:
        if ch in "([{":
            depth += 1
        elif ch in ")]}":
            depth -= 1
        if ch == "," and depth == 0:
            out.append(cur.strip())
            cur = ""
        else:
            cur += ch
    if cur.strip():
        out.append(cur.strip())
    return out


def _c_declarations():
    text = open(HEADER).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    text = re.sub(r"//[^\n]*", "", text)
    decls = {}
    for m in re.finditer(r"\b(fpx_[a-z_0-9]+)\s*\(([^;{}]*?)\)\s*;", text, flags=re.S):
        params = _split_top_level(m.group(2))
        decls[m.group(1)] = 0 if params == ["void"] else len(params)
    return decls


def _zig_block():
    md = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```zig\n(.*?)```", md, flags=re.S)
    assert blocks, "INTEGRATION.md has no zig block"
    return max(blocks, key=len)


def _zig_externs(block):
    text = re.sub(r"//[^\n]*", "", block)
    return [(m.group(1), len(_split_top_level(m.group(2)))) for m in re.finditer(r"pub extern fn (\w+)\s*\((.*?)\)\s*[^;()]*;", text, flags=re.S)]


def test_every_extern_fn_is_declared_once_and_matches_the_header():
    decls = _c_declarations()
    assert len(decls) >= 60
    externs = _zig_externs(_zig_block())
    names = [n for n, _ in externs]
    assert len(names) >= 50
    dup = sorted({n for n in names if names.count(n) > 1})
    assert not dup, f"declared more than once in the Zig block: {dup}"
    for name, arity in externs:
        assert name in decls, f"{name}: not declared by include/fpx.h"
        assert arity == decls[name], f"{name}: {arity} parameters in the Zig block, {decls[name]} in include/fpx.h"


def test_every_header_function_is_bound():
    """a maintainer who copies the block gets the WHOLE boundary (fpx_synth_* and the measurement helpers are test/bench only)"""
    decls = _c_declarations()
    bound = {n for n, _ in _zig_externs(_zig_block())}
    unbound = sorted(n for n in decls if n not in bound and not n.startswith(("fpx_synth", "fpx_measure")))
    assert not unbound, f"declared by include/fpx.h, missing from the Zig block: {unbound}"


def test_opaque_types_and_structs_are_declared_once():
    block = re.sub(r"//[^\n]*", "", _zig_block())
    consts = re.findall(r"pub const (\w+)\s*=", block)
    dup = sorted({n for n in consts if consts.count(n) > 1})
    assert not dup, dup
    used = set(re.findall(r"\*(?:const )?([A-Z]\w+)", block))
    assert used <= set(consts), f"types used but never declared: {sorted(used - set(consts))}"


C_CALLER = r"""
#include <stdio.h>
#include <string.h>
#include "fpx.h"

/* a C99 caller of the boundary: what the cgo / Zig @cImport side sees.  No GPU here: the context must refuse loudly. */
int main(void)
{
    fpx_ctx *ctx = NULL;
    fpx_opts opts;
    fpx_result out[4];
    uint32_t n = 0;
    int rc;
    memset(&opts, 0, sizeof opts);
    opts.max_results = 4;
    opts.min_score_pct = 10;
    if (fpx_version() < 2) return 2;
    if (strcmp(fpx_strerror(FPX_OK), "ok") != 0) return 3;
    rc = fpx_ctx_create(0, &ctx);
    if (rc != FPX_OK) {
        printf("no device: %s\n", fpx_last_error());
        return fpx_last_error()[0] ? 0 : 4;
    }
    {   /* an index without segments answers a search with nothing (src/Index.zig:170-177 over an empty segment list) */
        fpx_snapshot *snap = NULL;
        uint32_t hashes[3] = {1u, 2u, 3u};
        rc = fpx_snapshot_create(ctx, NULL, 0, &snap);
        if (rc != FPX_OK) return 5;
        rc = fpx_search(snap, hashes, 3, &opts, 0, out, 4, &n, NULL);
        if (rc != FPX_OK || n != 0) return 6;
        fpx_snapshot_release(snap);
    }
    fpx_ctx_destroy(ctx);
    printf("searched an empty index\n");
    return 0;
}
"""


@pytest.mark.skipif(shutil.which("gcc") is None, reason="no gcc")
def test_header_is_c99_and_a_c_caller_links(tmp_path):
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-x", "c", HEADER], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    lib_dir = os.path.join(ROOT, "acoustid-index_amd")
    if not os.path.exists(os.path.join(lib_dir, "libfpx.so")):
        pytest.skip("libfpx.so not built")
    src = tmp_path / "caller.c"
    src.write_text(C_CALLER)
    exe = tmp_path / "caller"
    r = subprocess.run(["gcc", "-std=c99", "-pedantic", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"), str(src), "-o", str(exe),
                        "-L", lib_dir, "-lfpx", f"-Wl,-rpath,{lib_dir}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
    assert r.returncode == 0, r.stderr
    r = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
    assert "no device" in r.stdout or "searched an empty index" in r.stdout
