#!/bin/bash
# debugging aid: the rt3 CLI with 1 and 3 shards on one device, several times; reports differing bytes
cd raytracer-3_amd
A="-f ppm -W 200 -H 113 --scene weekend --spp 4 --depth 12 --seed 3"
./rt3 $A --gpus 1 /tmp/g1a.ppm > /dev/null; ./rt3 $A --gpus 1 /tmp/g1b.ppm > /dev/null
for k in 1 2 3; do RT3_DEVICE_LIST=0,0,0 ./rt3 $A --gpus 3 /tmp/g3_$k.ppm > /dev/null; done
cmp -l /tmp/g1a.ppm /tmp/g1b.ppm | wc -l
for k in 1 2 3; do echo "run $k:"; cmp -l /tmp/g1a.ppm /tmp/g3_$k.ppm | head -20; done
