#!/bin/bash
# same-box A/B of two BUILDS of the library: tools/ab_two_libs.sh <old.so> <command...>   (runs the command with the tree's library,
# then with <old.so> in its place, then restores)
old=$1; shift
cp rii_amd/librii_amd.so /tmp/_new.so
echo "== new"; "$@"
cp "$old" rii_amd/librii_amd.so
echo "== old"; "$@"
cp /tmp/_new.so rii_amd/librii_amd.so
echo "== new again"; "$@"
