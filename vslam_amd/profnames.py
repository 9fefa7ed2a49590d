"""Kernel names as the profilers print them -> this repo's kernel identifiers and per-kernel timing slots.

rocprofv3 prints demangled names ("void (anonymous namespace)::min_eigen_tiered_kernel<true>(...)"); every kernel of the
library lives in an anonymous namespace at the top level and ends in `_kernel`, torch's live under `at::native::...`.
Shared by tools/prof_summary.py, tools/pmc_summary.py, tools/sq_summary.py and bench.py so that a new kernel shows up in
the committed summaries without anybody editing a name filter (round 3's `rbrief_tile_kernel` did not)."""
import re

_OURS = re.compile(r"^(?:void\s+)?(?:\(anonymous namespace\)::)?([a-z][a-z0-9_]*_kernel)\b")
# variants of one stage share the stage's timing slot (VsProfScope name) in the library's own per-kernel timing
_VARIANT = re.compile(r"_(v4|stream|tiered|lds|mfma|fp4|tile)_kernel$")
_SLOT = {"ransac_map_kernel": "ransac_sets_kernel", "match_spread_kernel": "match_knn2_kernel"}


def kernel_id(printed_name):
    """'min_eigen_tiered_kernel' for a kernel of this library, None for anything else (torch, runtime)."""
    m = _OURS.match(printed_name.strip().strip('"'))
    return m.group(1) if m else None


def slot_of(kid):
    """The library's timing slot (vslam_prof_*) a kernel identifier is accounted under."""
    return _SLOT.get(kid, _VARIANT.sub("_kernel", kid))
