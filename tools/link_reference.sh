#!/bin/bash
# link_reference.sh -- build a real MP-Gadget binary with the shim in the link, in a COPY of a reference checkout.
#
#   tools/link_reference.sh <reference checkout> <work dir> [--check]
#
# Needs what the reference's own build needs (an MPI compiler wrapper, GSL, and the depends/ libraries its Makefile builds: PFFT +
# FFTW, bigfile) plus ROCm for libmpgadget_hip.so.  None of GSL / PFFT / FFTW exists in the image this repository is developed in and
# there is no network, so this script has NEVER BEEN RUN TO COMPLETION there; `--check` (what tests/test_abi.py runs) stops after
# step 2 and only verifies that every file the steps name exists and that the shim files parse against the reference headers
# (gcc -fsyntax-only, with throw-away typedef stand-ins for <pfft.h> / <gsl/gsl_interp.h> in a temporary directory).
# What it does, so that a maintainer can follow it by hand (INTEGRATION.md explains each step):
#   1. copy the checkout (the reference tree itself is never modified);
#   2. copy shim/*.c, shim/*.h and include/mpgadget_hip.h into <copy>/libgadget/;
#   3. patch <copy>/libgadget/Makefile: drop gravpm.o gravshort-tree.o gravshort-pair.o gravity.o from GADGET_OBJS and add the shim
#      objects; rename the five tree constructors in forcetree.o and the eight integrator entry points in timestep.o / drift.o
#      (-Dname=cpu_name); guard the three SPH loops of density.c / hydra.c (-DMPGADGET_HIP);
#   4. add the two parameter hooks (set_densitypar, set_hydro_params), the accessors of mpg_shim.h and the
#      mpg_shim_particles_changed() calls listed in INTEGRATION.md ("Where P[] is reordered") with sed;
#   5. build the library of this repository, then `make` in the copy with LIBS += -L<repo>/mp-gadget_amd -lmpgadget_hip -lmpi.
set -eu
REF=${1:?usage: link_reference.sh <reference checkout> <work dir> [--check]}
WORK=${2:?usage: link_reference.sh <reference checkout> <work dir> [--check]}
CHECK=${3:-}
ROOT=$(cd "$(dirname "$0")/.." && pwd)
SHIM_C="gravity-hip.c sph-hip.c forcetree-hip.c timestep-hip.c mpg_mpi_comm.c mpg_rccl_mpi.c"
SHIM_H="mpg_shim.h mpg_shim_epoch.h mpg_mpi_comm.h"
REF_FILES="libgadget/Makefile libgadget/gravpm.c libgadget/gravshort-tree.c libgadget/gravshort-pair.c libgadget/gravity.c libgadget/forcetree.c
           libgadget/density.c libgadget/hydra.c libgadget/timestep.c libgadget/drift.c libgadget/timebinmgr.c libgadget/run.c libgadget/domain.c
           libgadget/exchange.c libgadget/fof.c libgadget/slotsmanager.c gadget/Makefile Makefile.rules Options.mk.example"

echo "== 0. inputs"
for f in $REF_FILES; do test -f "$REF/$f" || { echo "missing in the reference checkout: $f"; exit 1; }; done
for f in $SHIM_C $SHIM_H; do test -f "$ROOT/shim/$f" || { echo "missing shim file: $f"; exit 1; }; done
test -f "$ROOT/include/mpgadget_hip.h"

echo "== 1. copy of the checkout -> $WORK/MP-Gadget"
mkdir -p "$WORK"
if [ "$CHECK" = "--check" ]; then
    mkdir -p "$WORK/MP-Gadget"
    (cd "$REF" && tar cf - libgadget gadget Makefile.rules Options.mk.example Makefile Makefile.version 2>/dev/null) | (cd "$WORK/MP-Gadget" && tar xf -)
else
    cp -a "$REF" "$WORK/MP-Gadget"
fi
T="$WORK/MP-Gadget"

echo "== 2. the shim into libgadget/"
for f in $SHIM_C $SHIM_H; do cp "$ROOT/shim/$f" "$T/libgadget/"; done
cp "$ROOT/include/mpgadget_hip.h" "$T/libgadget/"
if [ "$CHECK" = "--check" ]; then
    STUB=$(mktemp -d)
    mkdir -p "$STUB/gsl"
    MPIINC=${MPI_INCLUDE:-/opt/conda/include}
    printf '#include <stddef.h>\n#include <mpi.h>\ntypedef double pfft_complex[2];\ntypedef struct pfft_plan_s *pfft_plan;\n' > "$STUB/pfft.h"
    printf 'typedef struct gsl_interp gsl_interp;\ntypedef struct gsl_interp_accel gsl_interp_accel;\n' > "$STUB/gsl/gsl_interp.h"
    for f in $SHIM_C; do
        gcc -std=gnu11 -fopenmp -fsyntax-only -Wall -Wextra -Werror -I "$STUB" -I "$MPIINC" -I "$T/libgadget" -I "$T" "$T/libgadget/$f" \
            || { echo "the shim file $f does not parse against this checkout"; exit 1; }
    done
fi

echo "== 3. libgadget/Makefile"
MK="$T/libgadget/Makefile"
sed -i -e 's/\bgravshort-tree\.o gravshort-pair\.o hydra\.o/hydra.o/' -e 's/\bgravpm\.o powerspectrum\.o/powerspectrum.o/' \
       -e 's/\bpetapm\.o gravity\.o/petapm.o/' "$MK"
sed -i -e 's/^\(GADGET_OBJS =  \\\)$/\1\n\t gravity-hip.o sph-hip.o forcetree-hip.o timestep-hip.o mpg_mpi_comm.o mpg_rccl_mpi.o \\/' "$MK"
cat >> "$MK" <<MKEOF

# ---- MP-Gadget on libmpgadget_hip (tools/link_reference.sh)
CFLAGS += -DMPGADGET_HIP -I$ROOT/include
.objs/forcetree.o: CFLAGS += -Dforce_tree_full=cpu_force_tree_full -Dforce_tree_rebuild_mask=cpu_force_tree_rebuild_mask -Dforce_tree_active_moments=cpu_force_tree_active_moments -Dforce_tree_calc_moments=cpu_force_tree_calc_moments -Dforce_tree_free=cpu_force_tree_free
.objs/timestep.o: CFLAGS += -Dapply_half_kick=cpu_apply_half_kick -Dapply_PM_half_kick=cpu_apply_PM_half_kick -Dfind_hydro_timesteps=cpu_find_hydro_timesteps -Dfind_timesteps=cpu_find_timesteps -Dapply_hydro_half_kick=cpu_apply_hydro_half_kick -Dhierarchical_gravity_and_timesteps=cpu_hierarchical_gravity_and_timesteps -Dhierarchical_gravity_accelerations=cpu_hierarchical_gravity_accelerations
.objs/drift.o: CFLAGS += -Ddrift_all_particles=cpu_drift_all_particles
MKEOF

echo "== 4. hooks in the reference sources (INTEGRATION.md lists them; each is one line)"
# the three SPH loops leave density.c / hydra.c (sph-hip.c defines them)
python3 - "$T" <<'PYEOF'
import re, sys
T = sys.argv[1]
def find_def(s, name):
    """(start of the definition incl. its return-type line, index of its opening brace, index just behind its closing brace)"""
    for m in re.finditer(r"\b%s\s*\(" % re.escape(name), s):
        i, depth = m.end(), 1
        while depth:                      # the matching parenthesis
            depth += {"(": 1, ")": -1}.get(s[i], 0)
            i += 1
        j = i
        while s[j] in " \t\n":
            j += 1
        if s[j] != "{":
            continue                      # a prototype or a call
        start = s.rfind("\n", 0, m.start()) + 1
        if s[start:m.start()].strip() == "":          # the return type stands on the line above
            start = s.rfind("\n", 0, start - 1) + 1
        k, depth = j + 1, 1
        while depth:
            depth += {"{": 1, "}": -1}.get(s[k], 0)
            k += 1
        return start, j, k
    raise SystemExit("no definition of %s found" % name)
def guard(path, names):
    s = open(path).read()
    for n in names:
        a, _, b = find_def(s, n)
        s = s[:a] + "#ifndef MPGADGET_HIP\n" + s[a:b] + "\n#endif\n" + s[b:]
    open(path, "w").write(s)
def at_entry(path, name, text):
    s = open(path).read()
    _, brace, _ = find_def(s, name)
    s = s[:brace + 1] + "\n    " + text + s[brace + 1:]
    open(path, "w").write(s)
def before_return(path, name, text):
    s = open(path).read()
    _, _, end = find_def(s, name)
    s = s[:end - 1] + "    " + text + "\n" + s[end - 1:]
    open(path, "w").write(s)
def append(path, text):
    open(path, "a").write("\n" + text + "\n")
# the three SPH loops leave density.c / hydra.c (sph-hip.c defines them); their parameters are handed over
guard(T + "/libgadget/density.c", ["density", "set_init_hsml"])
guard(T + "/libgadget/hydra.c", ["hydro_force"])
before_return(T + "/libgadget/density.c", "set_densitypar", "{ void mpg_shim_set_densitypar(const struct density_params *dp); mpg_shim_set_densitypar(&DensityParams); }")
before_return(T + "/libgadget/hydra.c", "set_hydro_params",
              "{ void mpg_shim_set_hydropar(int, double, double); mpg_shim_set_hydropar(HydroParams.DensityIndependentSphOn, HydroParams.DensityContrastLimit, HydroParams.ArtBulkViscConst); }")
# the accessors of mpg_shim.h next to the file-static parameters they read
append(T + "/libgadget/timestep.c", "double mpg_shim_max_gas_vel(void) { return TimestepParams.MaxGasVel; }\n"
       "double mpg_shim_min_size_timestep(void) { return TimestepParams.MinSizeTimestep; }\n"
       "double mpg_shim_courant_fac(void) { return TimestepParams.CourantFac; }\n"
       "double mpg_shim_err_tol_int_accuracy(void) { return TimestepParams.ErrTolIntAccuracy; }\n"
       "int mpg_shim_force_equal_timesteps(void) { return TimestepParams.ForceEqualTimesteps; }\n"
       "inttime_t mpg_shim_get_PM_timestep_ti(const DriftKickTimes *times, double atime, const Cosmology *CP, int FastParticleType, double asmth)\n"
       "{ return get_PM_timestep_ti(times, atime, CP, FastParticleType, asmth); }")
append(T + "/libgadget/timebinmgr.c", "#include <mpgadget_hip.h>\nvoid mpg_shim_timeline(mpg_timeline *tl)\n{\n    static double loga[8192];\n    int i;\n"
       "    for(i = 0; i < NSyncPoints && i < 8192; i++)\n        loga[i] = SyncPoints[i].loga;\n    tl->nsync = NSyncPoints;\n    tl->loga = loga;\n}")
# P[] is reordered / exchanged: the shim's upload cache must hear of it (the 64-record hash of mpg_shim_epoch.h is only a backstop)
hook = "{ extern void mpg_shim_particles_changed(void); mpg_shim_particles_changed(); }"
for path, func in (("/libgadget/domain.c", "domain_decompose_full"), ("/libgadget/domain.c", "domain_maintain"),
                   ("/libgadget/exchange.c", "domain_exchange"), ("/libgadget/slotsmanager.c", "slots_gc_sorted"),
                   ("/libgadget/slotsmanager.c", "slots_gc")):
    at_entry(T + path, func, hook)
print("hooks written")
PYEOF

if [ "$CHECK" = "--check" ]; then
    # the patched reference files and the renamed objects still parse (nothing is compiled to an object, linked or run)
    for f in density.c hydra.c timestep.c timebinmgr.c domain.c exchange.c slotsmanager.c drift.c forcetree.c; do
        extra=""
        case $f in
            forcetree.c) extra="-Dforce_tree_full=cpu_force_tree_full -Dforce_tree_rebuild_mask=cpu_force_tree_rebuild_mask -Dforce_tree_active_moments=cpu_force_tree_active_moments -Dforce_tree_calc_moments=cpu_force_tree_calc_moments -Dforce_tree_free=cpu_force_tree_free";;
            timestep.c) extra="-Dapply_half_kick=cpu_apply_half_kick -Dapply_PM_half_kick=cpu_apply_PM_half_kick -Dfind_hydro_timesteps=cpu_find_hydro_timesteps -Dfind_timesteps=cpu_find_timesteps -Dapply_hydro_half_kick=cpu_apply_hydro_half_kick -Dhierarchical_gravity_and_timesteps=cpu_hierarchical_gravity_and_timesteps -Dhierarchical_gravity_accelerations=cpu_hierarchical_gravity_accelerations";;
            drift.c) extra="-Ddrift_all_particles=cpu_drift_all_particles";;
        esac
        if ! gcc -std=gnu11 -fopenmp -fsyntax-only -DMPGADGET_HIP $extra -I "$STUB" -I "$MPIINC" -I "$ROOT/include" -I "$T/libgadget" -I "$T" "$T/libgadget/$f" 2> "$STUB/err"; then
            if grep -q 'gsl/.*No such file' "$STUB/err"; then
                echo "   ($f includes GSL headers this image lacks: not parsed here)"
            else
                cat "$STUB/err"; echo "the patched $f does not parse"; exit 1
            fi
        fi
    done
    grep -q "gravity-hip.o sph-hip.o forcetree-hip.o timestep-hip.o" "$MK" && ! grep -q "gravpm.o\|gravshort-tree.o\|gravity.o" "$MK" \
        || { echo "libgadget/Makefile was not patched as intended"; exit 1; }
    rm -rf "$STUB"
    echo "check passed: every file named exists; the shim, the patched reference files and the renamed objects parse (nothing was built)"
    exit 0
fi

echo "== 5. build"
python3 "$ROOT/mp-gadget_amd/build.py"
cp "$T/Options.mk.example" "$T/Options.mk"
echo "LIBS += -L$ROOT/mp-gadget_amd -lmpgadget_hip -Wl,-rpath,$ROOT/mp-gadget_amd" >> "$T/Options.mk"
(cd "$T" && make)
echo "built: $T/gadget/MP-Gadget"
