//! Replaces openvm/build.rs:11-20 for the HIP backend: instead of compiling `cuda/src/**/*.cu` with `CudaBuilder`,
//! build (or just link) `libpowdr_gpu.so` — same library name as build.rs:16 — from this repository's HIP sources.
//!
//! POWDR_GPU_LIB_DIR   directory holding a prebuilt libpowdr_gpu.so (powdr_amd/lib after `python -m powdr_amd.build`)
//! POWDR_AMD_ROOT      checkout of this repository: build the library here with hipcc (gfx950) if no prebuilt one is given
//! ROCM_PATH           default /opt/rocm
use std::{env, path::PathBuf, process::Command};

fn main() {
    println!("cargo:rerun-if-env-changed=POWDR_GPU_LIB_DIR");
    println!("cargo:rerun-if-env-changed=POWDR_AMD_ROOT");
    let rocm = env::var("ROCM_PATH").unwrap_or_else(|_| "/opt/rocm".into());
    let lib_dir = match env::var("POWDR_GPU_LIB_DIR") {
        Ok(dir) => PathBuf::from(dir),
        Err(_) => {
            let root = PathBuf::from(
                env::var("POWDR_AMD_ROOT").expect("set POWDR_GPU_LIB_DIR (prebuilt) or POWDR_AMD_ROOT (sources)"),
            );
            let out = PathBuf::from(env::var("OUT_DIR").unwrap());
            // one hipcc invocation per translation unit, like powdr_amd/build.py (gfx950 only: no multi-arch fat binary)
            let csrc = root.join("powdr_amd/csrc");
            let mut objects = vec![];
            let mut sources: Vec<PathBuf> = std::fs::read_dir(&csrc).unwrap().map(|e| e.unwrap().path()).collect();
            sources.extend(std::fs::read_dir(csrc.join("host")).unwrap().map(|e| e.unwrap().path()));
            for src in sources {
                let ext = src.extension().and_then(|e| e.to_str()).unwrap_or("");
                if ext != "hip" && ext != "cpp" {
                    continue;
                }
                println!("cargo:rerun-if-changed={}", src.display());
                let obj = out.join(src.file_stem().unwrap()).with_extension("o");
                let mut cmd = Command::new(format!("{rocm}/bin/hipcc"));
                cmd.args(["-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-I"])
                    .arg(root.join("include"));
                if ext == "cpp" {
                    cmd.args(["-x", "hip"]);
                }
                let ok = cmd.arg("-c").arg(&src).arg("-o").arg(&obj).status().expect("hipcc not found").success();
                assert!(ok, "hipcc failed on {}", src.display());
                objects.push(obj);
            }
            let ok = Command::new(format!("{rocm}/bin/hipcc"))
                .args(["-shared", "-fPIC", "--offload-arch=gfx950", "-o"])
                .arg(out.join("libpowdr_gpu.so"))
                .args(&objects)
                .status()
                .unwrap()
                .success();
            assert!(ok, "linking libpowdr_gpu.so failed");
            out
        }
    };
    println!("cargo:rustc-link-search=native={}", lib_dir.display());
    println!("cargo:rustc-link-search=native={rocm}/lib");
    println!("cargo:rustc-link-lib=dylib=powdr_gpu"); // openvm/build.rs:16 `.library_name("powdr_gpu")`
    println!("cargo:rustc-link-lib=dylib=amdhip64");
    // DEP_POWDR_GPU_INCLUDE for dependants, like CudaBuilder's DEP_*_INCLUDE (openvm/build.rs:12-13)
    if let Ok(root) = env::var("POWDR_AMD_ROOT") {
        println!("cargo:include={root}/include");
    }
}
