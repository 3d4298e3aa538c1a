#!/bin/bash
# Round 5, before the first GPU call (runs here, no GPU): the variant libraries scripts/r5/call1.sh compares.
#   trlead         - tracker.hip of branch next/tracker-leader (loads in front of stores in the leader's step / accept blocks)
#   trstamps       - main's tracker.hip with the device stamps (-DLDSO_STAMPS: "[tr stamps] ... solve / step / post" on stderr per track)
#   trlead_stamps  - the branch's tracker.hip with the stamps
set -e
cd "$(dirname "$0")/../.."
git show next/tracker-leader:ldso_amd/csrc/tracker.hip > /tmp/tracker_lead.hip
bash scripts/build_variant.sh trlead tracker.hip "" /tmp/tracker_lead.hip
bash scripts/build_variant.sh trstamps tracker.hip "-DLDSO_STAMPS"
bash scripts/build_variant.sh trlead_stamps tracker.hip "-DLDSO_STAMPS" /tmp/tracker_lead.hip
# the factorisation switches of branch next/solve-variants (ba_solve.hip): base = the branch with both switches off (= main's code, scheduled slightly differently)
git show next/solve-variants:ldso_amd/csrc/ba_solve.hip > /tmp/ba_solve_sv.hip
bash scripts/build_variant.sh sv_base ba_solve.hip "" /tmp/ba_solve_sv.hip
bash scripts/build_variant.sh sv_keeper3 ba_solve.hip "-DLD_KEEPER_WAVE=3" /tmp/ba_solve_sv.hip
bash scripts/build_variant.sh sv_skip ba_solve.hip "-DLD_P1_SKIP=1" /tmp/ba_solve_sv.hip
bash scripts/build_variant.sh sv_keeper3skip ba_solve.hip "-DLD_KEEPER_WAVE=3 -DLD_P1_SKIP=1" /tmp/ba_solve_sv.hip
bash scripts/build_variant.sh sv_keeper2skip ba_solve.hip "-DLD_KEEPER_WAVE=2 -DLD_P1_SKIP=1" /tmp/ba_solve_sv.hip
