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
