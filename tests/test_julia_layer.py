"""The Julia host layer (julia/SMMHip.jl, julia/SMMHipBackend.jl) cannot be executed here (no julia binary in the image).
What can be checked without one: the `struct` blocks mirror include/smmhip.h field for field (names, order, C offsets,
sizes — against a compiled probe of the header), every `ccall` names an exported symbol with the header's argument count,
the glue has the reference's field names and defines the methods SMM.jl dispatches on, and the block structure of both
files is balanced (a coarse syntax check)."""
import os
import re

import pytest
import subprocess
import tempfile

from smm_jl_amd import _abi as A

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
RAW = os.path.join(ROOT, "julia", "SMMHip.jl")
GLUE = os.path.join(ROOT, "julia", "SMMHipBackend.jl")
SHARDED = os.path.join(ROOT, "julia", "SMMHipSharded.jl")
HEADER = os.path.join(ROOT, "include", "smmhip.h")

JL2C = {"SmmProblem": "smm_problem_t", "SmmBgpOpts": "smm_bgp_opts_t", "SmmTables": "smm_tables_t",
        "SmmHistory": "smm_history_t", "SmmState": "smm_state_t", "SmmTiming": "smm_timing_t"}
# Julia type -> (size, alignment, C spelling as it appears in the header)
JLTYPES = {"Cint": (4, 4, "int32_t"), "Int32": (4, 4, "int32_t"), "Cdouble": (8, 8, "double"), "UInt64": (8, 8, "uint64_t"),
           "Int64": (8, 8, "int64_t"), "UInt8": (1, 1, "uint8_t"), "Int8": (1, 1, "int8_t")}
PTR_TARGET = {"Cdouble": "double", "Int32": "int32_t", "UInt8": "uint8_t", "Int8": "int8_t", "Cvoid": "void"}


def strip_julia(src):
    """drop comments, docstrings and string literals (keeps the code's block structure)"""
    src = re.sub(r'"""(?:.|\n)*?"""', '""', src)
    src = re.sub(r'"(?:\\.|[^"\\\n])*"', '""', src)
    src = re.sub(r"#=(?:.|\n)*?=#", "", src)
    return "\n".join(l.split("#", 1)[0] for l in src.splitlines())


def julia_structs(path):
    src = strip_julia(open(path).read())
    out = {}
    for m in re.finditer(r"^\s*(?:mutable\s+)?struct\s+(\w+)(?:\s*<:\s*[\w.]+)?\s*\n(.*?)^\s*end\b", src, re.S | re.M):
        fields = []
        for line in m.group(2).splitlines():
            for part in line.split(";"):
                part = part.strip()
                if not part:
                    continue
                fm = re.match(r"(\w+)\s*::\s*(.+)$", part)
                assert fm, (m.group(1), part)
                fields.append((fm.group(1), fm.group(2).strip()))
        out[m.group(1)] = fields
    return out


def header_structs():
    src = re.sub(r"/\*.*?\*/", "", open(HEADER).read(), flags=re.S)
    out = {}
    for m in re.finditer(r"typedef\s+struct\s*\{(.*?)\}\s*(\w+)\s*;", src, re.S):
        fields = []
        for decl in m.group(1).split(";"):
            decl = " ".join(decl.split())
            if not decl:
                continue
            dm = re.match(r"(?:const\s+)?(\w+)\s*(\*?)\s*(\w+)$", decl)
            assert dm, decl
            fields.append((dm.group(3), dm.group(1) + ("*" if dm.group(2) else "")))
        out[m.group(2)] = fields
    return out


def test_struct_mirrors_match_the_header_field_for_field():
    js, hs = julia_structs(RAW), header_structs()
    assert set(JL2C) <= set(js), "SMMHip.jl lacks a struct: %s" % (set(JL2C) - set(js))
    # offsets and sizes of the C side, from the compiler
    lines = ["#include <stdio.h>", "#include <stddef.h>", '#include "smmhip.h"', "int main(void) {"]
    for cname, fields in hs.items():
        lines.append('printf("%s %%zu\\n", sizeof(%s));' % (cname, cname))
        for f, _ in fields:
            lines.append('printf("%s.%s %%zu\\n", offsetof(%s, %s));' % (cname, f, cname, f))
    lines.append("return 0; }")
    with tempfile.TemporaryDirectory() as d:
        open(os.path.join(d, "p.c"), "w").write("\n".join(lines))
        subprocess.check_call(["gcc", "-I", os.path.join(ROOT, "include"), os.path.join(d, "p.c"), "-o", os.path.join(d, "p")])
        c = dict(l.rsplit(" ", 1) for l in subprocess.check_output([os.path.join(d, "p")]).decode().strip().splitlines())
    for jname, cname in JL2C.items():
        jf, hf = js[jname], hs[cname]
        assert [f for f, _ in jf] == [f for f, _ in hf], "%s: field names / order differ from %s" % (jname, cname)
        off = 0
        maxal = 1
        for (fname, jt), (_, ct) in zip(jf, hf):
            pm = re.match(r"Ptr\{(\w+)\}$", jt)
            if pm:
                size, al = 8, 8
                assert ct.endswith("*") and ct[:-1] == PTR_TARGET[pm.group(1)], "%s.%s: %s vs %s" % (jname, fname, jt, ct)
            else:
                assert jt in JLTYPES, "%s.%s: unknown Julia type %s" % (jname, fname, jt)
                size, al, cspell = JLTYPES[jt]
                assert ct == cspell, "%s.%s: %s vs %s" % (jname, fname, jt, ct)
            off = (off + al - 1) // al * al
            assert off == int(c["%s.%s" % (cname, fname)]), "%s.%s: offset %d, C has %s" % (jname, fname, off, c["%s.%s" % (cname, fname)])
            off += size
            maxal = max(maxal, al)
        assert (off + maxal - 1) // maxal * maxal == int(c[cname]), "%s: size" % jname


def test_every_ccall_names_an_exported_symbol_with_the_right_arity():
    table = {name: args for name, _, args in A.SYMBOLS}
    seen = set()
    for path in (RAW, GLUE):
        src = strip_julia(open(path).read())
        for m in re.finditer(r"ccall\(\s*(?:sym\(:(\w+)\)|Libdl\.dlsym\(LIB\[\],\s*:(\w+)\))\s*,\s*(\w+)\s*,\s*\(([^()]*(?:\{[^()]*\}[^()]*)*)\)", src):
            name = m.group(1) or m.group(2)
            args = [a for a in (x.strip() for x in m.group(4).split(",")) if a]
            assert name in table, "ccall of %s: not a symbol of include/smmhip.h" % name
            assert len(args) == len(table[name]), "ccall of %s passes %d arguments, the header declares %d" % (name, len(args), len(table[name]))
            seen.add(name)
    need = {"smm_abi_version", "smm_ctx_create", "smm_ctx_destroy", "smm_last_error", "smm_bgp_step", "smm_get_history", "smm_get_state",
            "smm_set_state", "smm_eval_batch", "smm_register_user_objective"}
    assert need <= seen, need - seen
    assert "ABI_VERSION = %d" % A.load().smm_abi_version() in open(RAW).read()


def test_glue_has_the_reference_shape():
    glue = strip_julia(open(GLUE).read())
    st = julia_structs(GLUE)["MAlgoBGPHip"]
    # the fields the reference's generic code touches, in the reference's order (MAlgoBGP, src/mopt/AlgoBGP.jl:497-503)
    assert [f for f, _ in st][:6] == ["m", "opts", "i", "chains", "anim", "dist_fun"]
    assert dict(st)["chains"] == "Vector{BGPChain}" and dict(st)["m"] == "MProb" and dict(st)["i"] == "Int"
    assert re.search(r"mutable struct MAlgoBGPHip <: MAlgo", glue)
    # the method SMM.jl dispatches on (src/mopt/AlgoAbstract.jl:45) and the readers / persistence of src/SMM.jl:31-57
    for sig in (r"function computeNextIteration!\(algo::MAlgoBGPHip\)", r"function run!\(algo::MAlgoBGPHip\)",
                r"summary\(algo::MAlgoBGPHip\)", r"save\(algo::MAlgoBGPHip, filename::AbstractString\)",
                r"function restart!\(algo::MAlgoBGPHip, extraIter::Int\)", r"function MAlgoBGPHip\(m::MProb, opts::Dict\)",
                r"function MAlgoBGPHip\(ref::MAlgoBGP;", r"function getproperty\(algo::MAlgoBGPHip, s::Symbol\)",
                r"function sync_chains!\(algo::MAlgoBGPHip\)"):
        assert re.search(sig, glue), sig
    imports = re.search(r"import SMM:([^\n]*\n[^\n]*)", glue).group(1)
    for name in ("computeNextIteration!", "run!", "summary", "save", "restart!", "MAlgo", "BGPChain", "Eval", "MProb"):
        assert name in imports, name
    # every BGPChain field the sync fills exists in the reference's struct (src/mopt/AlgoBGP.jl:42-60)
    for f in ("evals", "accepted", "exchanged", "best_val", "best_id", "curr_val", "iter", "sigma", "accept_rate"):
        assert re.search(r"\bc\.%s\b" % f, glue), f
    for f in ("value", "prob", "accepted", "status", "params", "simMoments"):      # Eval fields, src/mopt/Eval.jl
        assert re.search(r"\bev\.%s\b" % f, glue), f


def test_block_structure_is_balanced():
    """coarse syntax check: block openers at bracket depth 0 and `end`s balance, brackets balance"""
    openers = {"module", "struct", "function", "if", "for", "while", "begin", "let", "try", "do", "macro", "quote"}
    for path in (RAW, GLUE, SHARDED):
        src = strip_julia(open(path).read())
        depth, blocks = 0, 0
        for tok in re.findall(r"[A-Za-z_]\w*!?|[()\[\]{}]", src):
            if tok in "([{":
                depth += 1
            elif tok in ")]}":
                depth -= 1
                assert depth >= 0, path
            elif tok == "end":
                if depth == 0:      # a[end] / x[1:end] inside brackets is an index, not a block end
                    blocks -= 1
                    assert blocks >= 0, path
            elif tok in openers and depth == 0:
                blocks += 1
        assert depth == 0 and blocks == 0, (path, depth, blocks)


REF = "/root/reference/src"


@pytest.mark.skipif(not os.path.isdir(REF), reason="the reference sources are only present in the build container")
def test_every_smm_symbol_the_glue_uses_exists_in_the_reference():
    """the glue has never met a Julia parser (no julia binary in the image), so at least every name it takes from SMM must exist
    there: exported by src/SMM.jl:31-57, or defined (function / struct / const / one-line method) somewhere under src/"""
    glue = strip_julia(open(GLUE).read())
    imports = re.search(r"import SMM:((?:[^\n]*,\s*\n)*[^\n]*)", glue).group(1)
    names = {n.strip() for n in imports.replace("\n", " ").split(",") if n.strip()}
    names |= set(re.findall(r"\bSMM\.([A-Za-z_]\w*!?)", glue))
    assert {"computeNextIteration!", "BGPChain", "extendBGPChain!", "restart!", "objfunc_norm"} <= names
    src = ""
    for root, _, files in os.walk(REF):
        for f in files:
            if f.endswith(".jl"):
                src += open(os.path.join(root, f), errors="replace").read() + "\n"
    exported = set(re.findall(r"[A-Za-z_]\w*!?", " ".join(re.findall(r"^export\s+((?:[^\n]*,\s*\n)*[^\n]*)", src, re.M))))
    missing = []
    for n in sorted(names):
        e = re.escape(n)
        defined = re.search(r"(?m)^\s*(?:function\s+(?:SMM\.)?%s\s*[\({]|(?:mutable\s+)?struct\s+%s\b|abstract\s+type\s+%s\b|const\s+%s\b|%s\([^)]*\)\s*=)" % (e, e, e, e, e), src)
        via_using = n == "DataFrame" and re.search(r"(?m)^using DataFrames\b", src)     # (a binding SMM has through `using DataFrames`, src/SMM.jl:9)
        if not (n in exported or defined or via_using):
            missing.append(n)
    assert not missing, "names the glue takes from SMM that the reference neither exports nor defines: %s" % missing


def test_glue_refuses_animate():
    # AlgoBGP.jl:621-624: the reference's per-iteration animation hook cannot be served by lazily filled chains
    glue = open(GLUE).read()
    assert re.search(r'get\(opts, "animate", false\) == true\s*&&\s*\n?\s*throw\(ArgumentError', glue)


JULIA_STDLIB = {"Base", "Core", "Libdl", "Random", "Statistics", "LinearAlgebra", "Distributed", "Logging", "Test", "Printf", "Dates", "Serialization",
                "SharedArrays", "SparseArrays", "Sockets", "Mmap", "InteractiveUtils", "DelimitedFiles", "Pkg", "UUIDs"}
# [deps] of the reference's Project.toml (:6-26): what `using SMM` guarantees to be installed in the user's environment
SMM_DEPS = {"DataFrames", "DataFramesMeta", "Distributed", "Distributions", "Documenter", "FileIO", "GLM", "JLD2", "JSON", "LinearAlgebra", "Logging",
            "OrderedCollections", "PDMats", "Plots", "ProgressMeter", "Random", "Revise", "Statistics", "StatsPlots", "Test"}


def test_every_package_the_julia_files_load_is_available_next_to_smm():
    """`using DataStructures` failed at load time in SMM.jl's own environment (it depends on
    OrderedCollections, /root/reference/Project.toml:18, src/SMM.jl:10).  Every `using` / `import` of both files must name the
    standard library, a dependency of SMM.jl, SMM itself or this repository's own modules."""
    if os.path.isfile("/root/reference/Project.toml"):   # (the committed list above is the reference's [deps] block: keep it honest)
        deps = set(re.findall(r"(?m)^(\w+)\s*=\s*\"[0-9a-f-]{36}\"", open("/root/reference/Project.toml").read().split("[deps]")[1].split("[compat]")[0]))
        assert deps == SMM_DEPS, deps ^ SMM_DEPS
    own = {"SMM", "SMMHip", "SMMHipBackend"}
    for path in (RAW, GLUE, SHARDED):
        src = strip_julia(open(path).read())
        for m in re.finditer(r"(?m)^\s*(?:using|import)\s+([^\n]+)", src):
            for item in m.group(1).split(":")[0].split(","):
                pkg = item.strip().lstrip(".").split(".")[0]
                assert pkg in JULIA_STDLIB | SMM_DEPS | own, "%s loads %r: neither the standard library nor a dependency of SMM.jl" % (os.path.basename(path), pkg)


def test_sync_chains_returns_at_once_when_nothing_was_stepped():
    glue = strip_julia(open(GLUE).read())
    body = re.search(r"function sync_chains!\(algo::MAlgoBGPHip\)(.*?)\nend", glue, re.S).group(1)
    early = body.index("getfield(algo, :stepped) == getfield(algo, :synced) && return chains")
    assert early < body.index("hip_state(hip)"), "the early return must come before anything that synchronises with the device"
    # ... and every path that enqueues iterations counts them
    assert len(re.findall(r"setfield!\(algo, :stepped,", glue)) >= 4
    assert os.path.isfile(os.path.join(ROOT, "julia", "FIRST_RUN.md"))


def test_single_steps_are_counted_and_enqueued_asynchronously():
    """VERDICT r4 "Next #6": the reference's unchanged run! loop (AlgoAbstract.jl:38-45) calls computeNextIteration! once per iteration;
    the glue must neither synchronise with the device nor make a library call per iteration (one iteration at a time never takes the
    persistent form): the calls are counted and flushed as ONE smm_bgp_step_async when somebody reads the chains"""
    glue = strip_julia(open(GLUE).read())
    body = re.search(r"function computeNextIteration!\(algo::MAlgoBGPHip\)(.*?)\nend", glue, re.S).group(1)
    assert "hip_step" not in body and "hip_sync" not in body and ":deferred" in body and "flush_steps!" in body
    flush = re.search(r"function flush_steps!\(algo::MAlgoBGPHip\)(.*?)\nend", glue, re.S).group(1)
    assert "hip_step_async!" in flush and "hip_sync" not in flush
    sync = re.search(r"function sync_chains!\(algo::MAlgoBGPHip\)(.*?)\nend", glue, re.S).group(1)
    assert sync.index("flush_steps!(algo)") < sync.index("getfield(algo, :stepped) == getfield(algo, :synced) && return chains") < sync.index("hip_sync(hip)")
    # a failing step still leaves the count at what the device completed
    assert len(re.findall(r"setfield!\(algo, :stepped, SMMHip\.hip_iter\(", glue)) >= 2
    raw = strip_julia(open(RAW).read())
    assert re.search(r"ccall\(sym\(:smm_bgp_step_async\)", raw)


def test_sharded_driver_uses_only_distributed_and_the_binding():
    """VERDICT r4 "Next #6": julia/SMMHipSharded.jl — a Distributed-only multi-GPU driver (no MPI.jl): every SMMHip function it calls is
    defined in the binding, the handles travel by remotecall_fetch, step / finish run under @sync (the barrier the header asks for)"""
    src = strip_julia(open(SHARDED).read())
    raw = strip_julia(open(RAW).read())
    glue = strip_julia(open(GLUE).read())
    used = set(re.findall(r"\bSMMHip\.(hip_\w+!?)", src))
    assert {"hip_p2p_init", "hip_p2p_attach!", "hip_p2p_step!", "hip_p2p_finish!", "hip_sync", "hip_history", "hip_state", "hip_destroy!"} <= used, used
    for f in used:
        assert re.search(r"(?m)^(?:function\s+)?%s\(" % re.escape(f), raw), "SMMHip.%s is not defined in the binding" % f
    assert re.search(r"function hip_context\(m::MProb, opts::Dict; N_local::Int", glue) and "SMMHipBackend.hip_context(" in src
    assert "using Distributed" in src and "MPI" not in src
    assert len(re.findall(r"@sync for", src)) >= 5 and "remotecall_fetch(shard_finish" in src and "remotecall_fetch(shard_attach" in src
    for name in ("smm_bgp_p2p_init", "smm_bgp_p2p_attach", "smm_bgp_p2p_step", "smm_bgp_p2p_finish", "smm_get_persistent", "smm_set_persistent"):
        assert re.search(r"ccall\(sym\(:%s\)" % name, raw), name
