#!/bin/bash
# run_ref.sh — run the compiled reference (oracle/_ref/ref_render) on one scene.
# Authoring container only. The reference's XMLs carry the author's absolute
# macOS paths: a path-remapped COPY is written under oracle/_ref/.
#   usage: run_ref.sh <scene path relative to SceneFiles> <W> <H> <tag> [threads] [spp] [paths]   (spp: recipe S; "paths": recipe P)
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")/../.." && pwd)
OUT=$HERE/oracle/_ref/out/$4
mkdir -p "$OUT"
case "$1" in
  /*) # a scene directory of this repository (tests/scenes/...): instantiate its @DIR@ placeholders in a copy
      SRC=$(dirname "$1"); rm -rf "$OUT/scene_dir"; mkdir -p "$OUT/scene_dir"
      for f in "$SRC"/*; do
        case "$f" in *.xml|*.mtl|*.obj) sed "s#@DIR@#$OUT/scene_dir#g" "$f" > "$OUT/scene_dir/$(basename "$f")";; *) cp "$f" "$OUT/scene_dir/";; esac
      done
      cp "$OUT/scene_dir/$(basename "$1")" "$OUT/scene.xml";;
  *)  sed "s#/Users/Peter/GitRepos/RayTracer-Utah#$REF#g" "$REF/SceneFiles/$1" > "$OUT/scene.xml";;
esac
MODE=""
if [ -n "$6" ]; then if [ -n "$7" ]; then MODE="--paths $6"; else MODE="--spp $6"; fi; fi
"$HERE/oracle/_ref/ref_render" "$OUT/scene.xml" "$2" "$3" "$OUT" "${5:-8}" $MODE | tail -1
