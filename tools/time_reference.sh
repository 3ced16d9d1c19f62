#!/bin/sh
# Builds and times the real reference (the Rust crate streamz-rs) on this machine's host cores,
# if and only if a Rust toolchain and an offline crate registry are present; otherwise says
# that it was skipped.  usage: tools/time_reference.sh /path/to/StreamZ/streamz-rs train_files.txt
#
# The list file names 16-bit 44.1 kHz WAV clips with labels, one "path,label" per line, as the
# reference's CLI reads them (main.rs:41-79).  The reference prints nothing machine-readable,
# so the figure is wall-clock seconds for the whole run; windows = sum over files of
# (samples - 800) / 400 + 1 (lib.rs:288-291) and the CLI trains 100 epochs (main.rs:33).
set -eu
crate="${1:-}"
list="${2:-}"
if [ -z "$crate" ] || [ ! -f "$crate/Cargo.toml" ]; then
    echo "skipped: no reference checkout given (usage: $0 <streamz-rs dir> <train_files.txt>)"
    exit 0
fi
if ! command -v cargo >/dev/null 2>&1; then
    echo "skipped: cargo not found"
    exit 0
fi
if ! (cd "$crate" && cargo fetch --offline >/dev/null 2>&1); then
    echo "skipped: the crates of Cargo.lock are not available offline"
    exit 0
fi
(cd "$crate" && RUSTFLAGS="-C target-cpu=native" cargo build --release --offline)
bin="$crate/target/release/StreamZ"
[ -x "$bin" ] || bin=$(find "$crate/target/release" -maxdepth 1 -type f -perm -u+x | head -n 1)
work=$(mktemp -d)
cp "$list" "$work/train_files.txt"
echo "cores: $(nproc)"
start=$(date +%s.%N)
(cd "$work" && "$bin")
end=$(date +%s.%N)
echo "reference wall-clock seconds: $(echo "$end - $start" | bc)"
