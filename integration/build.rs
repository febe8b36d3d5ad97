// build.rs — links rabitq-rs against librbq.so (the MI355X candidate-scan engine, include/rbq.h) when the `gpu`
// feature is on.  Add to Cargo.toml:
//     [features]
//     gpu = []
//     [package]
//     build = "build.rs"
// and point RBQ_LIB_DIR at the directory that holds librbq.so (rabitq-rs_amd/csrc of this repository after
// `python -c "import __graft_entry__ as g; g.build()"`).  Never compiled in this repository's container (no cargo).
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=RBQ_LIB_DIR");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    if env::var_os("CARGO_FEATURE_GPU").is_none() {
        return;
    }
    let dir = match env::var_os("RBQ_LIB_DIR") {
        Some(d) => PathBuf::from(d),
        None => panic!("feature `gpu`: set RBQ_LIB_DIR to the directory that contains librbq.so"),
    };
    if !dir.join("librbq.so").exists() {
        panic!("feature `gpu`: {} holds no librbq.so", dir.display());
    }
    println!("cargo:rustc-link-search=native={}", dir.display());
    println!("cargo:rustc-link-lib=dylib=rbq");
    // librbq.so needs the HIP runtime it was built against (libamdhip64.so.7)
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string());
    println!("cargo:rustc-link-search=native={}/lib", rocm);
    // run-time lookup without LD_LIBRARY_PATH
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", dir.display());
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}/lib", rocm);
    // A host that issues batches from several threads or opens replicas on one device wants more than the default
    // four hardware queues: set GPU_MAX_HW_QUEUES=16 in the process environment before the first HIP call
    // (INTEGRATION.md G).
}
