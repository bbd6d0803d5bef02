#!/usr/bin/env python3
"""link_audit.py -- SYMBOL AUDIT of the drop-in: are the shim files link-complete against the reference's own objects?

    python tools/link_audit.py <reference checkout> <work dir> [--json out.json]

This is NOT a build of MP-Gadget (the image has neither GSL nor PFFT / FFTW) and NOT a parity pin: nothing is run.  It answers the three
link-time questions the parse-only checks (tests/test_abi.py) cannot:
  1. do shim/*.c COMPILE to object files against the reference's real headers (gcc -c, -Wall -Wextra -Werror)?
  2. with gravpm.o gravshort-tree.o gravshort-pair.o gravity.o out of the link, the three SPH loops guarded out of density.o / hydra.o and the
     renames of forcetree.o / timestep.o / drift.o applied (tools/link_reference.sh steps 1-4 on a scratch copy), is any symbol DEFINED TWICE
     across {shim objects, reference objects, libmpgadget_hip.so}?
  3. which symbols stay UNRESOLVED once every object that compiles here is on the table - and are they all accounted for: pfft_* / fftw_* / gsl_*
     (the libraries this image lacks), MPI, OpenMP / libc / libm, or a definition found in the source of one of the reference files that
     cannot be compiled here because it includes GSL / PFFT / FFTW headers (listed by file)?
What is compiled: every libgadget/*.c and utils/*.c of the patched scratch copy that gets through gcc -c with the two typedef-only stand-ins
for <pfft.h> and <gsl/gsl_interp.h> the parse test also uses plus two of the same kind for <gsl/gsl_interp2d.h> and <gsl/gsl_integration.h>
(names of types only, no function, no constant - a file that CALLS GSL therefore still fails on the implicit declaration and stays out);
bigfile's own sources for big_file_*.  Files that need more than that are left uncompiled and only searched for definitions by name."""
import json
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REMOVED = {"gravpm.c", "gravshort-tree.c", "gravshort-pair.c", "gravity.c"}          # replaced by shim/gravity-hip.c
RENAMES = {
    "forcetree.c": ["force_tree_full", "force_tree_rebuild_mask", "force_tree_active_moments", "force_tree_calc_moments", "force_tree_free"],
    "timestep.c": ["apply_half_kick", "apply_PM_half_kick", "find_hydro_timesteps", "find_timesteps", "apply_hydro_half_kick",
                   "hierarchical_gravity_and_timesteps", "hierarchical_gravity_accelerations"],
    "drift.c": ["drift_all_particles"],
}
SHIM_C = ["gravity-hip.c", "sph-hip.c", "forcetree-hip.c", "timestep-hip.c", "mpg_mpi_comm.c", "mpg_rccl_mpi.c"]
EXTERNAL = [("pfft", r"^pfft_"), ("fftw", r"^fftw_"), ("gsl", r"^gsl_"), ("mpi", r"^P?MPI_"), ("openmp", r"^(GOMP_|omp_)"),
            ("hdf5", r"^H5")]


def sh(cmd, **kw):
    return subprocess.run(cmd, capture_output=True, text=True, **kw)


def nm(path, dynamic=False):
    """(defined strong symbols, weak / common symbols, undefined symbols) of an object or shared library"""
    out = sh(["nm", "-D", path] if dynamic else ["nm", path]).stdout
    strong, weak, undef = set(), set(), set()
    for line in out.splitlines():
        parts = line.split()
        if len(parts) == 2:
            kind, name = parts
        elif len(parts) == 3:
            _, kind, name = parts
        else:
            continue
        if kind == "U":
            undef.add(name)
        elif kind in "TDBRSG":
            strong.add(name)
        elif kind in "WVC" or kind in "wv":
            weak.add(name)
    return strong, weak, undef


def libc_symbols():
    syms = set()
    for lib in ("libc.so.6", "libm.so.6", "libpthread.so.0", "libgomp.so.1", "libdl.so.2"):
        for d in ("/lib/x86_64-linux-gnu", "/usr/lib/x86_64-linux-gnu", "/lib64", "/usr/lib64"):
            p = os.path.join(d, lib)
            if os.path.exists(p):
                out = sh(["nm", "-D", "--defined-only", p]).stdout
                syms |= {l.split()[-1].split("@")[0] for l in out.splitlines() if l.split()}
                break
    return syms


def audit(ref, work):
    mpi_inc = os.environ.get("MPI_INCLUDE", "/opt/conda/include")
    r = sh([os.path.join(ROOT, "tools", "link_reference.sh"), ref, work, "--check"])
    if r.returncode != 0:
        raise SystemExit("link_reference.sh --check failed:\n" + r.stdout[-2000:] + r.stderr[-2000:])
    T = os.path.join(work, "MP-Gadget")
    lg = os.path.join(T, "libgadget")
    stub = os.path.join(work, "stub")
    obj = os.path.join(work, "obj")
    os.makedirs(os.path.join(stub, "gsl"), exist_ok=True)
    os.makedirs(obj, exist_ok=True)
    open(os.path.join(stub, "pfft.h"), "w").write("#include <stddef.h>\n#include <mpi.h>\ntypedef double pfft_complex[2];\ntypedef struct pfft_plan_s *pfft_plan;\n")
    open(os.path.join(stub, "gsl", "gsl_interp.h"), "w").write("typedef struct gsl_interp gsl_interp;\ntypedef struct gsl_interp_accel gsl_interp_accel;\n")
    # two more of the same kind (a type NAME each, used as pointer members / parameters by metal_return.h, which run.c includes): without
    # them run.c - the caller whose references this audit is about - does not get through the front end
    open(os.path.join(stub, "gsl", "gsl_interp2d.h"), "w").write("typedef struct gsl_interp2d gsl_interp2d;\n")
    open(os.path.join(stub, "gsl", "gsl_integration.h"), "w").write("typedef struct gsl_integration_workspace gsl_integration_workspace;\n")
    bigfile_src = os.path.join(ref, "depends", "bigfile", "src")
    inc = ["-I", stub, "-I", mpi_inc, "-I", os.path.join(ROOT, "include"), "-I", lg, "-I", T, "-I", bigfile_src]
    base = ["gcc", "-std=gnu11", "-fopenmp", "-O0", "-c", "-DMPGADGET_HIP", "-Werror=implicit-function-declaration"]
    report = {"shim_objects": [], "reference_objects": [], "not_compiled": {}, "duplicates": [], "unresolved": {}, "unaccounted": []}
    objs = {}
    # 1. the shim: real object files, warnings are errors
    for f in SHIM_C:
        o = os.path.join(obj, "shim_" + f.replace(".c", ".o"))
        r = sh(base + ["-Wall", "-Wextra", "-Werror"] + inc + [os.path.join(lg, f), "-o", o])
        if r.returncode != 0:
            raise SystemExit("shim file %s does not compile:\n%s" % (f, r.stderr[-3000:]))
        objs["shim/" + f] = o
        report["shim_objects"].append(f)
    # 2. the reference objects that stay in the link
    srcs = sorted(f for f in os.listdir(lg) if f.endswith(".c") and f not in REMOVED and f not in SHIM_C)
    srcs += sorted("utils/" + f for f in os.listdir(os.path.join(lg, "utils")) if f.endswith(".c"))
    srcs = [f for f in srcs if not os.path.basename(f).startswith("test_")]
    for f in srcs:
        extra = ["-D%s=cpu_%s" % (n, n) for n in RENAMES.get(f, [])]
        o = os.path.join(obj, f.replace("/", "_").replace(".c", ".o"))
        r = sh(base + extra + inc + [os.path.join(lg, f), "-o", o])
        if r.returncode == 0:
            objs[f] = o
            report["reference_objects"].append(f)
        else:
            m = re.search(r"fatal error: (\S+): No such file", r.stderr) or re.search(r"error: [\W]*(\w+)[\W]* undeclared", r.stderr) or \
                re.search(r"error: implicit declaration of function [\W]*(\w+)", r.stderr)
            report["not_compiled"][f] = m.group(1) if m else "?"
    if os.path.isdir(bigfile_src):
        for f in sorted(os.listdir(bigfile_src)):
            if f.endswith(".c"):
                o = os.path.join(obj, "bigfile_" + f.replace(".c", ".o"))
                if sh(["gcc", "-std=gnu11", "-O0", "-c", "-I", bigfile_src, "-I", mpi_inc, os.path.join(bigfile_src, f), "-o", o]).returncode == 0:
                    objs["depends/bigfile/" + f] = o
    # 3. symbols
    lib = os.path.join(ROOT, "mp-gadget_amd", "libmpgadget_hip.so")
    lib_def = nm(lib, dynamic=True)[0] | nm(lib, dynamic=True)[1]
    defined, where, undef = set(), {}, {}
    for name, o in objs.items():
        s, w, u = nm(o)
        for x in s:
            if x in where:
                report["duplicates"].append((x, where[x], name))
            where[x] = name
        defined |= s | w
        for x in u:
            undef.setdefault(x, []).append(name)
    for x in sorted(defined & lib_def):
        if x.startswith("mpg_") and x in where:
            report["duplicates"].append((x, where[x], "libmpgadget_hip.so"))
    libc = libc_symbols()
    left = {x: v for x, v in undef.items() if x not in defined and x not in lib_def}
    # definitions by name in the sources that could not be compiled here
    nc_src = {f: open(os.path.join(lg, f), errors="replace").read() for f in report["not_compiled"]}
    for x in sorted(left):
        cls = None
        for k, pat in EXTERNAL:
            if re.match(pat, x):
                cls = k
        if cls is None and (x in libc or x.startswith("__")):
            cls = "libc/libm/libgomp"
        if cls is None and x == "_GLOBAL_OFFSET_TABLE_":
            cls = "the linker"
        if cls is None and x in ("GADGET_COMPILER_SETTINGS", "GADGET_VERSION"):
            cls = "config.c (written by the reference's Makefile: makeconfig.sh)"
        if cls is None:
            for f, src in nc_src.items():
                renamed = ["cpu_" + n for n in RENAMES.get(f, [])]
                nm_ = x[4:] if x in renamed else x
                if re.search(r"^(?:[A-Za-z_][\w \*]*?[ \*])?%s\s*\([^;{]*\)\s*\{" % re.escape(nm_), src, flags=re.M | re.S) or \
                   re.search(r"^[A-Za-z_][\w \*\[\]]*\b%s\b\s*(?:\[[^\]]*\])?\s*(?:=|;)" % re.escape(x), src, flags=re.M):
                    cls = "defined in " + f + " (not compiled here: needs " + report["not_compiled"][f] + ")"
                    break
        if cls is None:
            report["unaccounted"].append((x, left[x]))
        else:
            report["unresolved"].setdefault(cls, []).append(x)
    return report


if __name__ == "__main__":
    if len(sys.argv) < 3:
        raise SystemExit(__doc__)
    rep = audit(sys.argv[1], sys.argv[2])
    if "--json" in sys.argv:
        json.dump(rep, open(sys.argv[sys.argv.index("--json") + 1], "w"), indent=1)
    print("shim objects compiled (-Wall -Wextra -Werror): %s" % " ".join(rep["shim_objects"]))
    print("reference objects compiled: %d; not compiled here: %d" % (len(rep["reference_objects"]), len(rep["not_compiled"])))
    for f, why in sorted(rep["not_compiled"].items()):
        print("   %-28s needs %s" % (f, why))
    print("symbols defined twice: %s" % (rep["duplicates"] or "none"))
    for cls, names in sorted(rep["unresolved"].items()):
        print("unresolved, %s: %d  %s" % (cls, len(names), " ".join(names[:12]) + (" ..." if len(names) > 12 else "")))
    print("unresolved and UNACCOUNTED: %s" % (rep["unaccounted"] or "none"))
    sys.exit(1 if rep["duplicates"] or rep["unaccounted"] else 0)
