// The generation loop of the reference's src/bin/chat.rs:188-222 on token ids at temperature 0.
//   cargo run --release --example chat_greedy -- model.lmrs 32 1 15043 3186
use std::fs::File;

use lmrs_hip::transformer::Transformer;
use memmap2::Mmap;

fn main() {
    let a: Vec<String> = std::env::args().collect();
    if a.len() < 4 {
        eprintln!("usage: {} model.lmrs n_new token_id...", a[0]);
        std::process::exit(2);
    }
    let file = File::open(&a[1]).expect("Model file not found!");
    let data = unsafe { Mmap::map(&file).expect("mmap failed") };
    let (mut model, _) = Transformer::new(&data);
    let n_new: u32 = a[2].parse().expect("n_new");
    let prompt: Vec<u32> = a[3..].iter().map(|s| s.parse().expect("token id")).collect();
    for t in model.generate_greedy(&prompt, n_new, 0) {
        print!("{} ", t);
    }
    println!();
}
