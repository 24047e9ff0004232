// build.rs — compiles the HIP kernels + C ABI for gfx950 with hipcc and links them into the crate.
use std::{env, path::PathBuf, process::Command};

fn main() {
    let root = PathBuf::from(env::var("CARGO_MANIFEST_DIR").unwrap()).join("..");
    let csrc = root.join("rustracer_amd").join("csrc");
    let out = PathBuf::from(env::var("OUT_DIR").unwrap());
    let hipcc = env::var("HIPCC").unwrap_or_else(|_| "/opt/rocm/bin/hipcc".into());
    let lib = out.join("librthip.so");
    let status = Command::new(&hipcc)
        .args(&["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-fno-slp-vectorize", "-mllvm", "-amdgpu-sched-strategy=max-ilp", "-mllvm", "-structurizecfg-skip-uniform-regions=1", "-mllvm", "-enable-gvn-hoist", "-shared", "-o"])
        .arg(&lib)
        .arg(csrc.join("rt_kernels.hip"))
        .arg(csrc.join("rthip_api.cpp"))
        .arg(csrc.join("rt_bvh.cpp"))
        .arg(csrc.join("rthip_group.cpp"))
        .arg("-ldl") // RCCL is dlopen'ed by the first multi-device group
        .status()
        .expect("failed to run hipcc");
    assert!(status.success(), "hipcc failed");
    println!("cargo:rustc-link-search=native={}", out.display());
    println!("cargo:rustc-link-lib=dylib=rthip");
    for f in &["rt_kernels.hip", "rthip_api.cpp", "rt_bvh.cpp", "rthip_group.cpp", "rt_device.h", "rt_rng.h", "rt_bvh.h"] {
        println!("cargo:rerun-if-changed={}", csrc.join(f).display());
    }
    println!("cargo:rerun-if-changed={}", root.join("include").join("rthip.h").display());
}
