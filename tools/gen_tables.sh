#!/bin/sh
# Regenerates audio_codec_amd/csrc/lc3_tables.h from the reference's constants (needs /root/reference).
set -e
REF=${REF:-/root/reference/LC3plus_ETSI_src_v17171_20200723}
FL=$REF/src/floating_point
HERE=$(cd "$(dirname "$0")" && pwd)
TMP=$(mktemp -d)
gcc -std=c99 -O0 -w -I"$FL" -o "$TMP/gen" "$HERE/gen_tables.c" "$FL/constants.c"
"$TMP/gen" > "$HERE/../audio_codec_amd/csrc/lc3_tables.h"
rm -rf "$TMP"
echo "wrote audio_codec_amd/csrc/lc3_tables.h"
