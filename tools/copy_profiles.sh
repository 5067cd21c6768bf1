#!/bin/bash
# gpurun_out/summ_TAG/* (tools/profile_round.sh TAG) -> profiles/TAG_* and profiles/traffic.json
TAG=${1:?tag}
S=gpurun_out/summ_$TAG
for f in $S/*; do cp $f profiles/${TAG}_$(basename $f); done
python tools/make_traffic_json.py $S $TAG > profiles/traffic.json
rm -f profiles/${TAG}_traffic.json
ls profiles | grep "^${TAG}_" | wc -l
