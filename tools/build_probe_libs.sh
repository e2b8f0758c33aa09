#!/bin/bash
# Builds probe variants of the library next to the product build:
#   tools/build_probe_libs.sh name:-Dflag[,-Dflag...] ...
# The bound-finding builds of the bf16 contraction name a K-loop policy from tools/probe/v2_policies.h:
#   tools/build_probe_libs.sh nomfma:-include,tools/probe/v2_policies.h,-DSKF_V2_POLICY=V2NoMfma \
#                             nodma:-include,tools/probe/v2_policies.h,-DSKF_V2_POLICY=V2NoDma ...
cd "$(dirname "$0")/.."
for spec in "$@"; do
  name=${spec%%:*}; flags=${spec#*:}; [ "$flags" = "$spec" ] && flags=""
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -I include ${flags//,/ } scikit-fusion_amd/csrc/skf_api.hip -o scikit-fusion_amd/lib/libskf_$name.so &
done
wait
ls -la scikit-fusion_amd/lib/
