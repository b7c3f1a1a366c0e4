//! Dumps, from the sbr crate's PUBLIC API (and from `rand` 0.5 directly), everything the MI355X engine's reference-order mode
//! can be compared with: the generator streams the index path rests on, the MovieLens protocol's split, and for each of the
//! five protocol cases of the crate's own tests (src/models/lstm.rs:450-520, src/models/ewma.rs:463-507) the returned loss,
//! the test MRR and the per-user test ranks; plus a REPLAY of the training driver's index work (src/models/
//! sequence_model.rs:76-98, :109, :137) outside the crate — subsequence order after the shuffle, partition seeds, the first
//! epoch's visiting order, the worker's raw negative draws — which needs the state of the hyper-parameter generator after
//! `build()`.  That state is private; it is reproduced by replaying `build_params`' draws on a CLONE of the generator:
//! exact for EWMA (every draw is in the crate's own source: src/models/ewma.rs:33-41, 167-198), and for the LSTM under the
//! stated assumption that wyrm's `nn::lstm::Parameters::new(d, d, rng)` draws four [2d x d] normal matrices
//! (`assumed_wyrm_lstm_draws` in the output; if wyrm draws differently the comparison tool says where the streams part).
//!
//! No file of the crate is copied here; the few lines that restate private helpers cite them.
//! NOTE: written without a Rust toolchain at hand — never compiled by its author.  Expect to fix a borrow or an import.

use std::env;
use std::fs::File;
use std::io::Write;

use rand::distributions::{Distribution, Normal, Uniform};
use rand::{Rng, SeedableRng, XorShiftRng};
use serde::Serialize;

use sbr::data::{user_based_split, CompressedInteractions, Interaction, Interactions};
use sbr::evaluation::mrr_score;
use sbr::models::{ewma, lstm, Loss, Optimizer, Parallelism};
use sbr::OnlineRankingModel;

const SEED: [u8; 16] = [42; 16];
const MAX_LEN: usize = 128;
const DIM: usize = 32;
const EPOCHS: usize = 10;
const KEEP: usize = 1000;

#[derive(Serialize)]
struct Streams {
    /// XorShiftRng::from_seed([42; 16]).next_u32() x 8
    next_u32: Vec<u32>,
    /// Uniform::new(0, u64::MAX).sample x 2 from a fresh generator (the split keys, src/data.rs:77-78)
    uniform_u64: Vec<u64>,
    /// Uniform::new(0, 1683usize).sample x 16 from a fresh generator
    uniform_usize_1683: Vec<usize>,
    /// (0..10).collect(), shuffled by a fresh generator (Rng::shuffle)
    shuffle_10: Vec<usize>,
    /// Normal::new(0.0, 1.0).sample x 8 from a fresh generator (f64 bit patterns)
    normal_bits: Vec<u64>,
    /// fresh_generator.gen::<[u8; 16]>()
    gen_seed16: Vec<u8>,
}

#[derive(Serialize)]
struct Split {
    train_users_with_data: usize,
    test_users_with_data: usize,
    train_interactions: usize,
    test_interactions: usize,
    /// FNV-1a over the train CSR's item ids in (user, time) order
    train_items_fnv: u64,
    test_items_fnv: u64,
}

#[derive(Serialize)]
struct Replay {
    /// what was assumed about draws hidden in wyrm (0 for EWMA: nothing assumed)
    assumed_wyrm_lstm_draws: usize,
    num_subsequences: usize,
    /// after parameters.rng().shuffle(&mut subsequences): (first item id, length) of the first KEEP subsequences
    shuffled_order: Vec<(usize, usize)>,
    /// per worker: XorShiftRng::from_seed(parameters.rng().gen()) — the 16 seed bytes
    worker_seeds: Vec<Vec<u8>>,
    /// per worker: its partition after the FIRST epoch's thread_rng.shuffle — (first item id, length) of the first KEEP
    first_epoch_order: Vec<Vec<(usize, usize)>>,
    /// per worker: the next KEEP raw draws negative_item_range.sample(thread_rng) after that shuffle.  Hinge / BPR: draw t is the
    /// negative of the worker's t-th loss term.  WARP: a term consumes 1..5 of them (the engine's TRIES say how many).
    first_epoch_raw_draws: Vec<Vec<usize>>,
}

#[derive(Serialize)]
struct Case {
    name: String,
    fit_loss: f32,
    test_mrr: f32,
    train_mrr: f32,
    /// rank of the held-out item for every test user with >= 2 interactions, in user order (src/evaluation.rs:20-43)
    test_ranks: Vec<usize>,
    replay: Replay,
}

#[derive(Serialize)]
struct Dump {
    crate_name: &'static str,
    streams: Streams,
    split: Split,
    cases: Vec<Case>,
}

fn fnv(items: impl Iterator<Item = usize>) -> u64 {
    let mut h: u64 = 0xcbf29ce484222325;
    for x in items {
        for b in (x as u64).to_le_bytes().iter() {
            h ^= *b as u64;
            h = h.wrapping_mul(0x100000001b3);
        }
    }
    h
}

fn streams() -> Streams {
    let mut a = XorShiftRng::from_seed(SEED);
    let next_u32 = (0..8).map(|_| a.next_u32()).collect();
    let mut b = XorShiftRng::from_seed(SEED);
    let r = Uniform::new(0, std::u64::MAX);
    let uniform_u64 = vec![r.sample(&mut b), r.sample(&mut b)];
    let mut c = XorShiftRng::from_seed(SEED);
    let ru = Uniform::new(0, 1683usize);
    let uniform_usize_1683 = (0..16).map(|_| ru.sample(&mut c)).collect();
    let mut d = XorShiftRng::from_seed(SEED);
    let mut v: Vec<usize> = (0..10).collect();
    d.shuffle(&mut v);
    let mut e = XorShiftRng::from_seed(SEED);
    let n = Normal::new(0.0, 1.0);
    let normal_bits = (0..8).map(|_| n.sample(&mut e).to_bits()).collect();
    let mut f = XorShiftRng::from_seed(SEED);
    let s: [u8; 16] = f.gen();
    Streams { next_u32, uniform_u64, uniform_usize_1683, shuffle_10: v, normal_bits, gen_seed16: s.to_vec() }
}

fn csr_stats(m: &CompressedInteractions) -> (usize, usize, u64) {
    let users = m.iter_users().filter(|u| !u.is_empty()).count();
    let n: usize = m.iter_users().map(|u| u.len()).sum();
    let h = fnv(m.iter_users().flat_map(|u| u.item_ids.iter().cloned().collect::<Vec<_>>()));
    (users, n, h)
}

/// ranks through the public trait only (what src/evaluation.rs:20-43 computes before taking reciprocals)
fn test_ranks<T: OnlineRankingModel>(model: &T, test: &CompressedInteractions) -> Vec<usize> {
    let item_ids: Vec<usize> = (0..test.num_items()).collect();
    let mut out = Vec::new();
    for user in test.iter_users().filter(|u| u.item_ids.len() >= 2) {
        let train_items = &user.item_ids[..user.item_ids.len() - 1];
        let test_item = *user.item_ids.last().unwrap();
        let rep = model.user_representation(train_items).unwrap();
        let mut p = model.predict(&rep, &item_ids).unwrap();
        for &t in train_items {
            p[t] = std::f32::MIN;
        }
        let s = p[test_item];
        out.push(p.iter().filter(|&&x| x >= s).count());
    }
    out
}

/// src/models/sequence_model.rs:76-98, :109, :137 replayed on `rng` = the hyper-parameter generator as `build()` left it
fn replay(train: &CompressedInteractions, mut rng: XorShiftRng, threads: usize, assumed: usize) -> Replay {
    let negative_item_range = Uniform::new(0, train.num_items());
    let mut subsequences: Vec<&[usize]> = train
        .iter_users()
        .flat_map(|user| user.chunks(MAX_LEN).map(|(item_ids, _)| item_ids).filter(|x| x.len() > 2).collect::<Vec<_>>())
        .collect();
    rng.shuffle(&mut subsequences);
    let shuffled_order = subsequences.iter().take(KEEP).map(|s| (s[0], s.len())).collect();
    let num_chunks = subsequences.len() / threads;
    let num_subsequences = subsequences.len();
    let mut worker_seeds = Vec::new();
    let mut first_epoch_order = Vec::new();
    let mut first_epoch_raw_draws = Vec::new();
    // the zip with the `threads` synchronised optimisers (:94-98) keeps the first `threads` chunks
    for chunk in subsequences.chunks_mut(num_chunks).take(threads) {
        let seed: [u8; 16] = rng.gen();
        let mut thread_rng = XorShiftRng::from_seed(seed);
        worker_seeds.push(seed.to_vec());
        thread_rng.shuffle(chunk);
        first_epoch_order.push(chunk.iter().take(KEEP).map(|s| (s[0], s.len())).collect());
        first_epoch_raw_draws.push((0..KEEP).map(|_| negative_item_range.sample(&mut thread_rng)).collect());
    }
    Replay { assumed_wyrm_lstm_draws: assumed, num_subsequences, shuffled_order, worker_seeds, first_epoch_order, first_epoch_raw_draws }
}

/// src/models/ewma.rs:33-41, 167-198 on a clone: embedding_init (num_items x dim normals), then fc1 and fc2 (dim x dim each)
fn advance_like_ewma_build(rng: &mut XorShiftRng, num_items: usize) {
    let e = Normal::new(0.0, 1.0 / DIM as f64);
    for _ in 0..num_items * DIM {
        let _ = e.sample(rng) as f32;
    }
    let d = Normal::new(0.0, (2.0 / (DIM + DIM) as f64).sqrt());
    for _ in 0..2 * DIM * DIM {
        let _ = d.sample(rng) as f32;
    }
}

/// src/models/lstm.rs:22-25, 174-194 on a clone: embedding_init, then — ASSUMED — wyrm's four [(DIM + DIM) x DIM] normal matrices
fn advance_like_lstm_build(rng: &mut XorShiftRng, num_items: usize) -> usize {
    let e = Normal::new(0.0, 1.0 / DIM as f64);
    for _ in 0..num_items * DIM {
        let _ = e.sample(rng) as f32;
    }
    let assumed = 4 * 2 * DIM * DIM;
    let w = Normal::new(0.0, 1.0 / ((2 * DIM) as f64).sqrt());
    for _ in 0..assumed {
        let _ = w.sample(rng) as f32;
    }
    assumed
}

fn main() {
    let args: Vec<String> = env::args().collect();
    if args.len() != 3 {
        eprintln!("usage: sbr_rust_check <data.csv of the sbr-rs repository> <output.json>");
        std::process::exit(2);
    }
    let mut reader = csv::Reader::from_path(&args[1]).expect("data.csv");
    let rows: Vec<Interaction> = reader.deserialize().collect::<Result<Vec<_>, _>>().expect("csv rows");
    let data = Interactions::from(rows);

    // run_test's set-up (src/models/lstm.rs:427-434): the SAME generator splits the data and then moves into the model
    let mut rng = XorShiftRng::from_seed(SEED);
    let (train, test) = user_based_split(&data, &mut rng, 0.2);
    let train_mat = train.to_compressed();
    let test_mat = test.to_compressed();
    let (tu, tn, th) = csr_stats(&train_mat);
    let (su, sn, sh) = csr_stats(&test_mat);
    let split = Split { train_users_with_data: tu, test_users_with_data: su, train_interactions: tn, test_interactions: sn, train_items_fnv: th, test_items_fnv: sh };

    let mut cases = Vec::new();
    let lstm_cases: [(&str, Loss, usize); 3] = [("lstm hinge 1 thread", Loss::Hinge, 1), ("lstm hinge 2 threads", Loss::Hinge, 2), ("lstm warp", Loss::WARP, 1)];
    for (name, loss, threads) in lstm_cases.iter().cloned() {
        let mut shadow = rng.clone();
        let assumed = advance_like_lstm_build(&mut shadow, data.num_items());
        let mut model = lstm::Hyperparameters::new(data.num_items(), MAX_LEN)
            .embedding_dim(DIM)
            .learning_rate(0.16)
            .l2_penalty(0.0004)
            .lstm_variant(lstm::LSTMVariant::Normal)
            .loss(loss)
            .optimizer(Optimizer::Adagrad)
            .parallelism(Parallelism::Synchronous)
            .num_epochs(EPOCHS)
            .num_threads(threads)
            .rng(rng.clone())
            .build();
        let fit_loss = model.fit(&train_mat).unwrap();
        cases.push(Case {
            name: name.to_string(),
            fit_loss,
            test_mrr: mrr_score(&model, &test_mat).unwrap(),
            train_mrr: mrr_score(&model, &train_mat).unwrap(),
            test_ranks: test_ranks(&model, &test_mat),
            replay: replay(&train_mat, shadow, threads, assumed),
        });
    }
    let ewma_cases: [(&str, Loss); 2] = [("ewma hinge", Loss::Hinge), ("ewma warp", Loss::WARP)];
    for (name, loss) in ewma_cases.iter().cloned() {
        let mut shadow = rng.clone();
        advance_like_ewma_build(&mut shadow, data.num_items());
        let mut model = ewma::Hyperparameters::new(data.num_items(), MAX_LEN)
            .embedding_dim(DIM)
            .learning_rate(0.16)
            .l2_penalty(0.0004)
            .loss(loss)
            .optimizer(Optimizer::Adagrad)
            .parallelism(Parallelism::Synchronous)
            .num_epochs(EPOCHS)
            .num_threads(1)
            .rng(rng.clone())
            .build();
        let fit_loss = model.fit(&train_mat).unwrap();
        cases.push(Case {
            name: name.to_string(),
            fit_loss,
            test_mrr: mrr_score(&model, &test_mat).unwrap(),
            train_mrr: mrr_score(&model, &train_mat).unwrap(),
            test_ranks: test_ranks(&model, &test_mat),
            replay: replay(&train_mat, shadow, 1, 0),
        });
    }
    let dump = Dump { crate_name: "sbr 0.5.0 (maciejkula/sbr-rs)", streams: streams(), split, cases };
    let mut f = File::create(&args[2]).expect("output");
    f.write_all(serde_json::to_string_pretty(&dump).unwrap().as_bytes()).unwrap();
    println!("wrote {}", &args[2]);
}
