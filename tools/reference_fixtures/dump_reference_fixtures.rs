// Dumps what the REAL reference computes for the inputs of this repository's golden fixtures, so
// that the CPU oracle and the HIP path can be pinned against it (tests/test_reference_fixtures.py).
//
// It cannot be built where this repository is developed (no Rust toolchain, no crates).  On a
// machine with cargo and the crates of streamz-rs/Cargo.lock:
//
//   python tools/reference_fixtures/make_inputs.py /tmp/szfix_in          # this repository
//   cp tools/reference_fixtures/dump_reference_fixtures.rs <StreamZ>/streamz-rs/examples/
//   cd <StreamZ>/streamz-rs
//   cargo run --release --example dump_reference_fixtures -- /tmp/szfix_in /tmp/szfix_out
//   cp /tmp/szfix_out/*.npy /tmp/szfix_out/*.npz <this repository>/tests/golden/reference/
//
// Outputs (all little-endian .npy / .npz as ndarray-npy writes them):
//   reference_mel.npy               (26, 401) f32  mel_filter::mel(44100, 800, Some(26), None, None, false, One)
//   reference_feat_<clip>.npy       (n, 60)  f32   FeatureExtractor::new().extract(pcm_<clip>)
//   reference_probs.npy             (rows, C) f32  SimpleNeuralNet::forward of every row of x.npy
//   reference_model_after.npz       the net after ONE train_batch(x, target, lr), via SimpleNeuralNet::save
use mel_filter::{mel, NormalizationFactor};
use ndarray::{Array1, Array2};
use ndarray_npy::{read_npy, write_npy};
use std::path::Path;
use streamz_rs::{FeatureExtractor, SimpleNeuralNet};

fn main() -> Result<(), Box<dyn std::error::Error>> {
    let args: Vec<String> = std::env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: dump_reference_fixtures <input dir> <output dir>");
        std::process::exit(2);
    }
    let (inp, out) = (Path::new(&args[1]), Path::new(&args[2]));
    std::fs::create_dir_all(out)?;

    // 1. the filterbank the extractor is built with (lib.rs:240-248)
    let filters: Vec<Vec<f32>> = mel::<f32>(44100, 800, Some(26), None, None, false, NormalizationFactor::One);
    let rows = filters.len();
    let cols = filters[0].len();
    let flat: Vec<f32> = filters.iter().flatten().copied().collect();
    write_npy(out.join("reference_mel.npy"), &Array2::from_shape_vec((rows, cols), flat)?)?;

    // 2. extract() of every pcm_<clip>.npy (int16, 1-D) listed in clips.txt
    let ext = FeatureExtractor::new();
    for clip in std::fs::read_to_string(inp.join("clips.txt"))?.lines() {
        let clip = clip.trim();
        if clip.is_empty() {
            continue;
        }
        let pcm: Array1<i16> = read_npy(inp.join(format!("pcm_{}.npy", clip)))?;
        let feats = ext.extract(pcm.as_slice().unwrap());
        let n = feats.len();
        let flat: Vec<f32> = feats.into_iter().flatten().collect();
        write_npy(out.join(format!("reference_feat_{}.npy", clip)), &Array2::from_shape_vec((n, 60), flat)?)?;
    }

    // 3. forward and one train_batch on the net of model_in.npz
    let mut net = SimpleNeuralNet::load(inp.join("model_in.npz").to_str().unwrap())?;
    let x: Array2<f32> = read_npy(inp.join("x.npy"))?;
    let target: Array1<f32> = read_npy(inp.join("target.npy"))?;
    let lr: Array1<f32> = read_npy(inp.join("lr.npy"))?;
    let batch: Vec<Vec<f32>> = x.rows().into_iter().map(|r| r.to_vec()).collect();
    let c = net.output_size();
    let mut probs = Vec::with_capacity(batch.len() * c);
    for row in &batch {
        probs.extend(net.forward(row));
    }
    write_npy(out.join("reference_probs.npy"), &Array2::from_shape_vec((batch.len(), c), probs)?)?;
    net.train_batch(&batch, target.as_slice().unwrap(), lr[0]);
    net.save(out.join("reference_model_after.npz").to_str().unwrap())?;
    Ok(())
}
