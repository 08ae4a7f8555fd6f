# Build a copy of the last commit under ab_prev/ (git-ignored, travels with gpurun) for same-box A/B timing:
#   bash tools/ab_setup.sh && gpurun -- 'bash tools/ab_bench.sh'
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
rm -rf "$root/ab_prev"
mkdir -p "$root/ab_prev"
git -C "$root" archive HEAD | tar -x -C "$root/ab_prev"
make -C "$root/ab_prev/maskplanner_amd/csrc" -s
rm -rf "$root/ab_prev/tests/golden" "$root/ab_prev/profiles" "$root/ab_prev/maskplanner_amd/lib/obj"
