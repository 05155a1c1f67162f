// Drop-in replacement for rust-api/lambda-snark-sys/build.rs that links the MI355X backend
// (liblambda_snark_core.so, built by `make -C lambda-snark-r_amd/csrc` of this repository) instead of building
// cpp-core with CMake and linking Microsoft SEAL / zstd / NTL / GMP / zlib.
//
// STATUS: written against the reference's build script (rust-api/lambda-snark-sys/build.rs:1-239) and checked, as far as
// this repository can check it without a Rust toolchain, by tests/test_build_rs_contract.py (every function the bindgen
// allow-list below lets through — lwe_.*, ntt_.*, lambda_snark_r1cs_.* — is declared in the headers named here AND exported
// by the shared library; struct layouts are pinned by tests/c/abi_conformance.c).  It has NOT been compiled by cargo: the
// build image has no rustc/cargo.
//
// What changed against the reference script, by its line numbers:
//   :25-29   rerun triggers   -> the backend's headers and library instead of ../../cpp-core, VCPKG_ROOT, SEAL_DIR
//   :31-104  vcpkg discovery + cmake::Config::new("../../cpp-core").build()   -> removed (nothing is compiled here; drop the
//            `cmake` build-dependency from Cargo.toml)
//   :106     static lambda_snark_core   -> dylib lambda_snark_core from $LAMBDA_SNARK_AMD_ROOT/lambda-snark-r_amd/lib
//   :119-180 seal-4.1, zstd, ntl, gmp, z and their Homebrew fallbacks   -> removed; amdhip64 from $ROCM_PATH/lib added
//   :171-180 pthread, m, stdc++   -> kept (the library is C++17)
//   :185-207 bindgen: same four header NAMES and the same allow-lists; the headers are plain C (no <NTL/ZZ_p.h>, no
//            `-x c++`, no hard-coded /usr/include/c++/14 paths); `lsr_.*` and `.*_batch` are allowed in addition
//   :209-233 NTL / SEAL include-path discovery   -> removed
//
// Environment:
//   LAMBDA_SNARK_AMD_ROOT   checkout of this repository (required)
//   ROCM_PATH               ROCm prefix, default /opt/rocm
//   LAMBDA_SNARK_AMD_BATCH  set to 0 to leave the additive batched / device entry points (batch.h, prover.h) out of the bindings
use std::env;
use std::path::PathBuf;

fn main() {
    println!("cargo:rerun-if-env-changed=LAMBDA_SNARK_AMD_ROOT");
    println!("cargo:rerun-if-env-changed=ROCM_PATH");
    println!("cargo:rerun-if-env-changed=LAMBDA_SNARK_AMD_BATCH");

    let root = PathBuf::from(env::var("LAMBDA_SNARK_AMD_ROOT").expect(
        "LAMBDA_SNARK_AMD_ROOT must point at the MI355X backend checkout (the directory holding include/ and lambda-snark-r_amd/)",
    ));
    let root = root.canonicalize().unwrap_or(root);
    let include = root.join("include");
    let lib_dir = root.join("lambda-snark-r_amd").join("lib");
    let library = lib_dir.join("liblambda_snark_core.so");
    if !library.exists() {
        panic!(
            "{} not found: build it first (python -c 'import __graft_entry__ as g; g.build()' or make -C lambda-snark-r_amd/csrc)",
            library.display()
        );
    }
    println!("cargo:rerun-if-changed={}", include.display());
    println!("cargo:rerun-if-changed={}", library.display());

    // the backend and the HIP runtime it needs
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-lib=dylib=lambda_snark_core");
    println!("cargo:rustc-link-arg=-Wl,-rpath,{}", lib_dir.display());
    let rocm = PathBuf::from(env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".to_string()));
    let rocm_lib = rocm.join("lib");
    if rocm_lib.exists() {
        println!("cargo:rustc-link-search=native={}", rocm_lib.display());
        println!("cargo:rustc-link-arg=-Wl,-rpath,{}", rocm_lib.display());
    }
    println!("cargo:rustc-link-lib=dylib=amdhip64");

    println!("cargo:rustc-link-lib=dylib=pthread");
    if cfg!(target_os = "linux") {
        println!("cargo:rustc-link-lib=dylib=m");
    }
    let target = env::var("TARGET").expect("TARGET not set by Cargo");
    if target.contains("linux") || target.contains("bsd") {
        println!("cargo:rustc-link-lib=stdc++");
    } else {
        panic!("the MI355X backend targets Linux + ROCm; TARGET = {}", target);
    }

    // Rust bindings from the same header names the reference binds (build.rs:185-189)
    let header = |name: &str| include.join("lambda_snark").join(name).to_string_lossy().to_string();
    let mut bindings = bindgen::Builder::default()
        .header(header("types.h"))
        .header(header("commitment.h"))
        .header(header("ntt.h"))
        .header(header("r1cs.h"))
        .clang_arg(format!("-I{}", include.display()))
        .parse_callbacks(Box::new(bindgen::CargoCallbacks::new()))
        .allowlist_function("lwe_.*")
        .allowlist_function("ntt_.*")
        .allowlist_function("lambda_snark_r1cs_.*")
        .allowlist_type("Lwe.*")
        .allowlist_type("Ntt.*")
        .allowlist_type("SparseMatrix")
        .allowlist_type("SparseEntry")
        .allowlist_type("R1CSWitness")
        .allowlist_type("PublicParams")
        .allowlist_type("ProfileType")
        .allowlist_type("LambdaSnarkError");
    if env::var("LAMBDA_SNARK_AMD_BATCH").map(|v| v != "0").unwrap_or(true) {
        bindings = bindings
            .header(header("batch.h"))
            .header(header("prover.h"))
            .allowlist_function("lsr_.*")
            .allowlist_function("sample_gaussian")
            .allowlist_type("Lsr.*");
    }
    let bindings = bindings.generate().expect("Unable to generate bindings");

    let out_path = PathBuf::from(env::var("OUT_DIR").unwrap());
    bindings
        .write_to_file(out_path.join("bindings.rs"))
        .expect("Couldn't write bindings!");
}
