# CIAOAlgorithmsAMD.jl -- Julia host side of the MI355X finite-sum path: the SVRG / SAGA / SAG / Finito constructors,
# functors, `iterator` and `solution` of kul-optec/CIAOAlgorithms.jl, with the bodies of the hot loops replaced by
# `ccall`s into libciao_hip.so (include/ciao_hip.h).  AMDGPU.jl is used ONLY for device/stream/buffer handles
# (ROCArray, its pointer, the HIP stream); no KernelAbstractions, no CUDA.jl.
#
# Covers every constructor of the reference: SVRG, SAGA, SAG, Finito (basic, LFinito, adaptive) and Proshi.
#
# STATUS: written against the C ABI and never executed -- the build image has no `julia` binary (DESIGN.md section 1).
# The always-runnable twin of this file is ciaoalgorithms.jl_amd/solvers.py, which binds the same symbols with ctypes
# and is what tests/ exercise.  Every `ccall` below has its ctypes counterpart in _lib.py (same order, same types).
module CIAOAlgorithmsAMD

using AMDGPU
using LinearAlgebra
using Printf
using Random
using ProximalOperators

export solution, solve_together

const RealOrComplex{R} = Union{R,Complex{R}}      # CIAOAlgorithms.jl:3
const Maybe{T} = Union{T,Nothing}                 # CIAOAlgorithms.jl:4

const libciao = get(ENV, "CIAO_HIP_LIB", "libciao_hip.so")

# ---- C structs (include/ciao_hip.h) ------------------------------------------------------------------------------------
struct CiaoProblem
    loss::Int32; dtype::Int32
    N::Int64; d::Int64; ld::Int64; N_total::Int64
    A::Ptr{Cvoid}; b::Ptr{Cvoid}
    lam::Float64
end
struct CiaoProxDesc
    kind::Int32; _pad::Int32
    lam::Float64; lo::Float64; hi::Float64
    lo_vec::Ptr{Cvoid}; hi_vec::Ptr{Cvoid}
end
const CIAO_MAX_SHARDS = 8
struct CiaoShardTable                       # ciao_shard_table: fixed-size C arrays are NTuples
    nshards::Int32; owner::Int32
    row0::NTuple{9,Int64}
    A::NTuple{8,Ptr{Cvoid}}; b::NTuple{8,Ptr{Cvoid}}; table::NTuple{8,Ptr{Cvoid}}; meta::NTuple{8,Ptr{Cvoid}}
end
const CIAO_ABI_VERSION = Int32(3)
const CIAO_F32, CIAO_F64 = Int32(0), Int32(1)
const LOSS_LS, LOSS_LOGISTIC, LOSS_ZERO, LOSS_LS_COMPLEX = Int32(0), Int32(1), Int32(2), Int32(3)
const PROX_ZERO, PROX_L1, PROX_BOX, PROX_L1_COMPLEX = Int32(0), Int32(1), Int32(2), Int32(3)
dtype_code(::Type{Float32}) = CIAO_F32
dtype_code(::Type{Float64}) = CIAO_F64

struct CiaoError <: Exception
    status::Int32
    msg::String
end
function check(status::Int32)
    status == 0 && return nothing
    throw(CiaoError(status, unsafe_string(ccall((:ciao_last_error, libciao), Cstring, ()))))
end

# ---- context: one device + the HIP stream AMDGPU.jl is using ----------------------------------------------------------
mutable struct Context
    h::Ptr{Cvoid}
    function Context(device::Integer = AMDGPU.device_id(AMDGPU.device()) - 1)
        v = ccall((:ciao_abi_version, libciao), Int32, ())
        v == CIAO_ABI_VERSION || error("libciao_hip.so has ABI version $v, this wrapper binds version $CIAO_ABI_VERSION")
        out = Ref{Ptr{Cvoid}}(C_NULL)
        stream = Base.unsafe_convert(Ptr{Cvoid}, AMDGPU.stream().stream)
        check(ccall((:ciao_ctx_create, libciao), Int32, (Int32, Ptr{Cvoid}, Ref{Ptr{Cvoid}}), device, stream, out))
        ctx = new(out[])
        finalizer(c -> ccall((:ciao_ctx_destroy, libciao), Int32, (Ptr{Cvoid},), c.h), ctx)
        return ctx
    end
end
synchronize(ctx::Context) = check(ccall((:ciao_ctx_synchronize, libciao), Int32, (Ptr{Cvoid},), ctx.h))
const default_ctx = Ref{Union{Nothing,Context}}(nothing)
context() = (default_ctx[] === nothing && (default_ctx[] = Context()); default_ctx[])

dptr(a::ROCArray) = Base.unsafe_convert(Ptr{Cvoid}, pointer(a))
dptr(::Nothing) = C_NULL

# complex T (CIAOAlgorithms.jl:3; test/test_lasso.jl:3 runs ComplexF32 / ComplexF64): every complex vector travels as its
# interleaved (re, im) pairs -- reinterpret(R, x) -- and d counts reals.  LeastSquares rows pair with Zero / NormL1 only.
# ---- feature padding (the Python mirror: solvers.py `_Iterable.__init__`, operators.pack_F(pad_to=...)) -------------------------------
# The fast chain kernels stream rows of whole 16-byte chunks from 16-byte aligned addresses; a row of, say, 1001 Float64 is neither and
# runs the register-ring chains at 2-3.7x the time per update.  A problem this wrapper packs ITSELF from host operators (not a PackedF
# the caller laid out, not a complex problem, not adaptive Finito) is packed with dp - d zero columns, dp = d rounded up to whole
# chunks: the extra coordinates multiply nothing and start at zero, so the d real ones evolve as without them (to rounding).  Every
# state vector is a length-d contiguous view of a dp-long device buffer -- what the caller sees keeps the reference's shapes and
# identities (`solution(state) === state.z`) -- and the library, told d = dp by the packed problem, reads and writes all dp entries.
const PAD_FEATURES = Ref(true)
padded_length(::Type{R}, d::Int) where {R} = (v = 16 ÷ sizeof(R); cld(d, v) * v)
# a length-d view at the head of a zeroed buffer of dp entries (AMDGPU.jl: a contiguous view of a ROCArray is a ROCArray on the same memory)
statevec(::Type{R}, d::Int, dp::Int) where {R} = view(AMDGPU.zeros(R, dp), 1:d)
function statevec(x::Vector{R}, dp::Int) where {R}
    buf = AMDGPU.zeros(R, dp)
    copyto!(buf, 1, x, 1, length(x))
    return view(buf, 1:length(x))
end

reals(::Type{R}, x::AbstractArray{<:Real}) where {R} = R.(vec(x))
reals(::Type{R}, x::AbstractArray{<:Complex}) where {R} = collect(reinterpret(R, Complex{R}.(vec(x))))
nreals(x0::AbstractArray) = eltype(x0) <: Complex ? 2 * length(x0) : length(x0)
host_solution(sol::ROCArray{R}, x0::AbstractArray) where {R} =
    eltype(x0) <: Complex ? reshape(collect(reinterpret(Complex{R}, Array(sol))), size(x0)) : reshape(Array(sol), size(x0))

# ---- packing: recognise the operator families of the reference's tests (SURVEY.md section 8b) -------------------------
# F::Vector of one-row LeastSquares / Precompose(LogisticLoss) / Zero  ->  (A as a d x N Julia matrix = row-major N x d)
struct PackedF{R}
    loss::Int32
    A::Union{Nothing,ROCArray{R,2}}     # d x N column-major  ==  N x d row-major, ld = d
    b::Union{Nothing,ROCArray{R,1}}
    lam::Float64
    N::Int
    d::Int
end
# F or g is not a family the device path packs.  The functors' `fallback = :host` catches exactly this and hands the problem
# to the REFERENCE's own loop (kul-optec/CIAOAlgorithms.jl: any ProximalOperators object, one operator call per sample, on the
# host) -- never silently: a warning names the route.  The reference package is an optional dependency: tell the wrapper where
# it is with `CIAOAlgorithmsAMD.use_reference!(CIAOAlgorithms)`.  (Python twin: solvers.py `_route`, host_route.py.)
struct UnpackableOperator <: Exception
    msg::String
end
Base.showerror(io::IO, e::UnpackableOperator) = print(io, "UnpackableOperator: ", e.msg)
const REFERENCE = Ref{Any}(nothing)
use_reference!(m::Module) = (REFERENCE[] = m; nothing)
function host_route(name::Symbol, ::Type{R}, cfg::NamedTuple, x0; kwargs...) where {R}
    REFERENCE[] === nothing &&
        throw(ArgumentError("fallback = :host needs the reference package: `using CIAOAlgorithms; CIAOAlgorithmsAMD.use_reference!(CIAOAlgorithms)`"))
    @warn "CIAOAlgorithmsAMD: this problem runs on the HOST route (the reference's own loop, one operator call per sample, no GPU): its F / g are not a family the device path packs"
    ref_solver = getfield(REFERENCE[], name){R}(; cfg...)      # CIAOAlgorithms.SVRG{R}(; γ, maxit, ...) etc.: the same keywords
    return ref_solver(x0; kwargs...)
end
# iterator(solver, x0; ...) on the device, or `nothing` when the problem is unpackable and the caller allowed the host route
# dp for a problem given as host operators: d itself for a PackedF the caller laid out, for complex problems and where the solver does
# not pad (adaptive Finito, ProShI keep their d: solvers.py `_pads_features`)
function padded_d(::Type{R}, F, d::Int, cplx::Bool; pads::Bool = true) where {R}
    (PAD_FEATURES[] && pads && !cplx && !(F isa PackedF) && d >= 1) ? padded_length(R, d) : d
end

function device_iterator(solver, x0, fallback; kwargs...)
    try
        return iterator(solver, x0; kwargs...)
    catch e
        (e isa UnpackableOperator && fallback === :host) || rethrow()
        return nothing
    end
end

function pack_F(::Type{R}, F, N::Int, d::Int; cplx::Bool = false, pad_to::Int = d) where {R}
    F isa PackedF{R} && return F
    dp = cplx ? d : pad_to           # rows of the packed matrix: d real columns + dp - d zero columns (feature padding, above)
    if F === nothing || all(f -> f isa ProximalOperators.Zero, F)
        return PackedF{R}(LOSS_ZERO, nothing, nothing, 0.0, N, dp)
    elseif cplx && all(f -> f isa ProximalOperators.LeastSquares, F)
        # complex x0: rows of d/2 complex entries (real rows are widened: real A times complex x is what LeastSquares computes)
        lam = F[1].lambda
        n = d ÷ 2
        all(f -> f.lambda == lam && size(f.A) == (1, n), F) || throw(UnpackableOperator("unpackable LeastSquares terms"))
        A = Matrix{Complex{R}}(undef, n, N); b = Vector{Complex{R}}(undef, N)
        for i in 1:N
            A[:, i] .= vec(F[i].A); b[i] = F[i].b[1]
        end
        return PackedF{R}(LOSS_LS_COMPLEX, ROCArray(reshape(collect(reinterpret(reshape, R, A)), d, N)),      # 2 x n x N -> (re, im) interleaved rows
                          ROCArray(collect(reinterpret(R, b))), Float64(lam), N, d)
    elseif cplx
        throw(UnpackableOperator("with a complex x0 the device path packs LeastSquares rows and Zero only"))
    elseif all(f -> f isa ProximalOperators.LeastSquares, F)
        lam = F[1].lambda
        all(f -> f.lambda == lam && size(f.A) == (1, d), F) || throw(UnpackableOperator("unpackable LeastSquares terms"))
        A = zeros(R, dp, N); b = Vector{R}(undef, N)
        for i in 1:N
            A[1:d, i] .= vec(F[i].A); b[i] = F[i].b[1]
        end
        return PackedF{R}(LOSS_LS, ROCArray(A), ROCArray(b), Float64(lam), N, dp)
    elseif all(f -> f isa ProximalOperators.Precompose && f.f isa ProximalOperators.LogisticLoss, F)
        A = zeros(R, dp, N); y = Vector{R}(undef, N)
        for i in 1:N
            size(F[i].L) == (1, d) || throw(UnpackableOperator("unpackable Precompose term"))
            A[1:d, i] .= vec(F[i].L); y[i] = F[i].f.y[1]
        end
        return PackedF{R}(LOSS_LOGISTIC, ROCArray(A), ROCArray(y), 1.0, N, dp)
    end
    throw(UnpackableOperator("F is not a family the device path can pack (LeastSquares rows, Precompose(LogisticLoss) rows, Zero)"))
end
cproblem(p::PackedF{R}) where {R} =
    CiaoProblem(p.loss, dtype_code(R), p.N, p.d, p.d, p.N, dptr(p.A), dptr(p.b), p.lam)

function pack_g(::Type{R}, g, d::Int; cplx::Bool = false, pad_to::Int = d) where {R}
    g isa ProximalOperators.Zero && return (CiaoProxDesc(PROX_ZERO, 0, 0.0, -Inf, Inf, C_NULL, C_NULL), nothing)
    g isa ProximalOperators.NormL1 && g.lambda isa Real &&
        return (CiaoProxDesc(cplx ? PROX_L1_COMPLEX : PROX_L1, 0, Float64(g.lambda), -Inf, Inf, C_NULL, C_NULL), nothing)
    cplx && throw(UnpackableOperator("with a complex x0 g must be Zero or NormL1 (IndBox has no complex form)"))
    if g isa ProximalOperators.IndBox
        # (per-coordinate bounds of a padded problem: the padding coordinates are unconstrained and stay at zero)
        lo = g.lb isa Real ? nothing : ROCArray(vcat(R.(vec(g.lb)), fill(R(-Inf), pad_to - d)))
        hi = g.ub isa Real ? nothing : ROCArray(vcat(R.(vec(g.ub)), fill(R(Inf), pad_to - d)))
        keep = (lo, hi)
        return (CiaoProxDesc(PROX_BOX, 0, 0.0, g.lb isa Real ? Float64(g.lb) : -Inf, g.ub isa Real ? Float64(g.ub) : Inf,
                             dptr(lo), dptr(hi)), keep)
    end
    throw(UnpackableOperator("g is not a family the device path supports (Zero, NormL1, IndBox)"))
end

# ---- L1 plugin API on packed operators (the ProximalOperators.jl calling convention, device arrays) ---------------------
# gradient!(y, F, i, x): y = ∇f_i(x), returns f_i(x)            call sites SVRG_basic.jl:60,74,75,89; SAGA_basic.jl:43,56
function gradient!(y::ROCArray{R,1}, F::PackedF{R}, i::Integer, x::ROCArray{R,1}) where {R}
    fval = ROCArray{R}(undef, 1)
    check(ccall((:ciao_gradient, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProblem}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(cproblem(F)), i - 1, dptr(x), dptr(y), dptr(fval)))
    return Array(fval)[1]
end
# prox!(y, g, x, γ): y = prox_{γ g}(x) for a packed g = pack_g(R, g, d)[1]         call sites SVRG_basic.jl:80; SAGA_basic.jl:48,64
function prox!(y::ROCArray{R,1}, g::CiaoProxDesc, x::ROCArray{R,1}, γ::Real) where {R}
    check(ccall((:ciao_prox, libciao), Int32, (Ptr{Cvoid}, Int32, Int64, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Ptr{Cvoid}),
                context().h, dtype_code(R), length(x), Ref(g), dptr(x), Float64(γ), dptr(y)))
    return y
end
# av = (1/N) Σ ∇f_i(x): the full-gradient sweep on its own (SVRG_basic.jl:58-63), and fused with a prox step
function full_gradient!(av::ROCArray{R,1}, F::PackedF{R}, x::ROCArray{R,1}) where {R}
    check(ccall((:ciao_full_gradient, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProblem}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(cproblem(F)), dptr(x), dptr(av)))
    return av
end
function proxgrad_step!(y::ROCArray{R,1}, av::ROCArray{R,1}, F::PackedF{R}, g::CiaoProxDesc, x::ROCArray{R,1}, γ::Real) where {R}
    check(ccall((:ciao_proxgrad_step, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(cproblem(F)), Ref(g), Float64(γ), dptr(x), dptr(av), dptr(y)))
    return y
end
# (1/N) Σ f_i(x) + g(x): what test/test_lasso.jl:45 computes on the host
function objective(F::PackedF{R}, g::CiaoProxDesc, x::ROCArray{R,1}) where {R}
    out = Ref{Float64}(0.0)
    check(ccall((:ciao_objective, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Ref{Float64}),
                context().h, Ref(cproblem(F)), Ref(g), dptr(x), out))
    return out[]
end
# multi-GPU: hand an RCCL communicator (ncclComm_t) to the context; tuning knobs
set_rccl!(comm::Ptr{Cvoid}, lib::AbstractString = "librccl.so") =
    check(ccall((:ciao_ctx_set_rccl, libciao), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Cstring), context().h, comm, lib))
# Row-sharded problem for the sequential chains (include/ciao_hip.h: ciao_ctx_set_shards): the chain owner reads the other
# ranks' rows through peer-mapped pointers (ipc_open of the handles the other processes made with ipc_export).
set_shards!(t::CiaoShardTable) =
    check(ccall((:ciao_ctx_set_shards, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoShardTable}), context().h, Ref(t)))
clear_shards!() = check(ccall((:ciao_ctx_set_shards, libciao), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), context().h, C_NULL))
function ipc_export(a::ROCArray)
    handle = zeros(UInt8, 64); off = Ref{Int64}(0)
    check(ccall((:ciao_ipc_export, libciao), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Ref{Int64}), dptr(a), handle, off))
    return handle, off[]
end
function ipc_open(handle::Vector{UInt8}, offset::Integer)
    out = Ref{Ptr{Cvoid}}(C_NULL)
    check(ccall((:ciao_ipc_open, libciao), Int32, (Ptr{UInt8}, Int64, Ref{Ptr{Cvoid}}), handle, offset, out))
    return out[]
end
ipc_close(p::Ptr{Cvoid}, offset::Integer) = check(ccall((:ciao_ipc_close, libciao), Int32, (Ptr{Cvoid}, Int64), p, offset))
# One-shot peer all-reduce (include/ciao_hip.h: ciao_ctx_set_peers; csrc/peer_kernels.h): every rank creates a mailbox, exchanges
# its IPC handle (ipc_export of the mailbox pointer / ipc_open, through whatever the host uses between processes: MPI.jl,
# Distributed ...), and sets the table; the d-vector sums then travel by the kernels' own stores, no collective call.
function peer_mailbox_create(max_elems::Integer)
    out = Ref{Ptr{Cvoid}}(C_NULL); nbytes = Ref{Int64}(0)
    check(ccall((:ciao_peer_mailbox_create, libciao), Int32, (Ptr{Cvoid}, Int64, Ref{Ptr{Cvoid}}, Ref{Int64}), context().h, max_elems, out, nbytes))
    return out[], nbytes[]
end
peer_mailbox_destroy(m::Ptr{Cvoid}) = check(ccall((:ciao_peer_mailbox_destroy, libciao), Int32, (Ptr{Cvoid}, Ptr{Cvoid}), context().h, m))
function ipc_export(p::Ptr{Cvoid})
    handle = zeros(UInt8, 64); off = Ref{Int64}(0)
    check(ccall((:ciao_ipc_export, libciao), Int32, (Ptr{Cvoid}, Ptr{UInt8}, Ref{Int64}), p, handle, off))
    return handle, off[]
end
set_peers!(rank::Integer, mailboxes::Vector{Ptr{Cvoid}}, max_elems::Integer) =      # rank is 0-based on the ABI
    check(ccall((:ciao_ctx_set_peers, libciao), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Ptr{Cvoid}}, Int64), context().h, rank, length(mailboxes), mailboxes, max_elems))
clear_peers!() = check(ccall((:ciao_ctx_set_peers, libciao), Int32, (Ptr{Cvoid}, Int32, Int32, Ptr{Ptr{Cvoid}}, Int64), context().h, 0, 0, C_NULL, 0))
# Several independent chains in ONE launch (include/ciao_hip.h: ciao_ctx_chain_batch_begin / _end): the `saga_steps!` calls made
# inside `chain_batch() do ... end` -- a regularisation path or folds: iterables over the same rows, a state each -- are recorded
# and run together, one workgroup (one compute unit) per chain; each state ends bitwise as if its call had been made alone.
# No counterpart in the reference, which solves one problem per call.
const BATCH_OPEN = Ref(false)
const BATCH_KEEP = Any[]          # the index arrays of the recorded calls: alive until the launch is enqueued
function chain_batch(f)
    check(ccall((:ciao_ctx_chain_batch_begin, libciao), Int32, (Ptr{Cvoid},), context().h))
    BATCH_OPEN[] = true
    ok = false
    try
        f()
        ok = true
    finally
        BATCH_OPEN[] = false
        st = ccall((:ciao_ctx_chain_batch_end, libciao), Int32, (Ptr{Cvoid}, Int32), context().h, Int32(ok))
        empty!(BATCH_KEEP)        # (frees are stream-ordered: behind the launch)
        ok && check(st)
    end
    return nothing
end
set_option!(key::AbstractString, value::Integer) =
    check(ccall((:ciao_ctx_set_option, libciao), Int32, (Ptr{Cvoid}, Cstring, Int64), context().h, key, value))
last_kernel() = unsafe_string(ccall((:ciao_ctx_last_kernel, libciao), Cstring, (Ptr{Cvoid},), context().h))

# Objective monitor (SURVEY.md 8f rank 4): `obj` is a ROCArray{Float64}(undef, 3) that every full pass fills with
# {F(x), (1/N) Σ f_i(x), g(x)}; `set_monitor!(nothing)` switches it off.
set_monitor!(g::CiaoProxDesc, obj::ROCArray{Float64,1}) =
    check(ccall((:ciao_ctx_set_monitor, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProxDesc}, Ptr{Cvoid}), context().h, Ref(g), dptr(obj)))
set_monitor!(::Nothing) =
    check(ccall((:ciao_ctx_set_monitor, libciao), Int32, (Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}), context().h, C_NULL, C_NULL))
build_flags() = unsafe_string(ccall((:ciao_build_flags, libciao), Cstring, ()))
const TRUST_SVRG_STATE = Ref(false)

# The reference draws from Julia's global RNG inside Base.iterate; here the draws are made on the host with the SAME
# calls (so a Julia user keeps the reference's sample stream) and shipped as 0-based Int64 device arrays.
to_dev_idx(idx::AbstractVector{<:Integer}) = ROCArray(Int64.(idx) .- 1)

# ======================================================================================================================
# SVRG  (src/algorithms/SVRG/SVRG.jl, SVRG_basic.jl)
# ======================================================================================================================
struct SVRG{R<:Real}
    γ::Maybe{R}; maxit::Int; verbose::Bool; freq::Int; m::Maybe{Int}; plus::Bool
    function SVRG{R}(; γ::Maybe{R} = nothing, maxit::Int = 10000, verbose::Bool = false, freq::Int = 1000,
                     m::Maybe{Int} = nothing, plus::Bool = false) where {R}
        @assert γ === nothing || γ > 0
        @assert maxit > 0
        @assert freq > 0
        new(γ, maxit, verbose, freq, m, plus)
    end
end
SVRG(::Type{R}; kwargs...) where {R} = SVRG{R}(; kwargs...)
SVRG(; kwargs...) = SVRG(Float64; kwargs...)

struct SVRG_basic_iterable{R<:Real,Tx}
    F::PackedF{R}; g::CiaoProxDesc; gkeep::Any; x0::Tx; N::Int
    L::Maybe{Union{Array{R},R}}; μ::Maybe{Union{Array{R},R}}; γ::Maybe{R}; m::Maybe{Int}; plus::Bool
end
mutable struct SVRG_basic_state{R<:Real}
    γ::R; m::Int
    av::ROCArray{R,1}; z::ROCArray{R,1}; z_full::ROCArray{R,1}; w::ROCArray{R,1}
end

function Base.iterate(iter::SVRG_basic_iterable{R}) where {R}          # SVRG_basic.jl:30-69
    N = iter.N
    m = iter.m === nothing ? N : iter.m
    if iter.γ === nothing
        if iter.plus
            @warn "provide a stepsize γ"; return nothing
        elseif iter.L === nothing || iter.μ === nothing
            @warn "smoothness or convexity parameter absent"; return nothing
        end
        L_M, μ_M = maximum(iter.L), maximum(iter.μ)
        γ = 1 / (10 * L_M)
        rho = (1 + 4 * L_M * γ^2 * μ_M * (N + 1)) / (μ_M * γ * N * (1 - 4L_M * γ))
        rho >= 1 && @warn "convergence condition violated...provide a stepsize!"
    else
        γ = iter.γ
    end
    xh = reals(R, iter.x0)
    d, dp = length(xh), iter.F.d                                   # dp > d: a feature-padded problem (statevec: length-d views of dp-long buffers)
    x0d = statevec(xh, dp)
    av, z, z_full, w = (statevec(R, d, dp) for _ in 1:4)
    p = Ref(cproblem(iter.F))
    check(ccall((:ciao_svrg_init, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, p, dptr(x0d), dptr(av), dptr(z), dptr(z_full), dptr(w)))
    state = SVRG_basic_state{R}(R(γ), m, av, z, z_full, w)
    return state, state
end

function Base.iterate(iter::SVRG_basic_iterable{R}, state::SVRG_basic_state{R}) where {R}   # SVRG_basic.jl:71-96
    idx = to_dev_idx(rand(1:iter.N, state.m))                                               # :73
    p, g = Ref(cproblem(iter.F)), Ref(iter.g)
    # reuse_rowdots: a Julia array has no in-place version counter, and `solution(state) === state.z_full` hands the
    # vector to user code, so the wrapper vouches for the cached a_i'z_full only when the user says the state is untouched
    # (TRUST_SVRG_STATE[] = true; include/ciao_hip.h: ciao_svrg_iterate).  The default recomputes both dot products.
    check(ccall((:ciao_svrg_iterate, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Int64, Ptr{Cvoid}, Int32, Int32,
                 Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, p, g, Float64(state.γ), state.m, dptr(idx), Int32(iter.plus), Int32(TRUST_SVRG_STATE[]),
                dptr(state.av), dptr(state.z), dptr(state.z_full), dptr(state.w)))
    iter.plus && (state.m *= 2)                                                             # :93
    return state, state
end
solution(state::SVRG_basic_state) = state.z_full                                            # SVRG_basic.jl:99

# av[k] = (1/N) sum_i grad f_i(x[k]) for K iterates in ONE pass over the rows (ciao_full_gradient_multi: matrix cores for
# d = 256 / 512 / 1024, K single sweeps otherwise).  Not in the reference (one problem per call): for hosts that advance several solves.
function full_gradient_multi!(avs::Vector{<:ROCArray{R}}, F::PackedF{R}, xs::Vector{<:ROCArray{R}}) where {R}
    length(avs) == length(xs) || throw(ArgumentError("as many outputs as iterates"))
    xt, at = Ptr{Cvoid}[dptr(x) for x in xs], Ptr{Cvoid}[dptr(a) for a in avs]
    p = Ref(cproblem(F))
    GC.@preserve xt at check(ccall((:ciao_full_gradient_multi, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProblem}, Int32, Ptr{Cvoid}, Ptr{Cvoid}),
                                   context().h, p, Int32(length(xs)), pointer(xt), pointer(at)))
    return avs
end

# K SVRG solves over the same rows (a regularisation path: iterables that differ in g) advanced by ONE reference iteration each:
# the K inner cycles (:73-82) recorded and launched as one chain batch -- one workgroup, one compute unit, per solve -- then every
# solve's epoch tail (:84-93).  Each state ends bitwise as Base.iterate(iter, state) leaves it with TRUST_SVRG_STATE[] = false.
# `one_pass = true` (all iterables over the same packed F, the same m): the K full passes of the epoch tails run as ONE pass over
# the rows with K right-hand sides on the matrix cores (ciao_svrg_epoch_tail_multi) -- to rounding, not bitwise, K Base.iterate's.
function iterate_together!(iters::Vector{SVRG_basic_iterable{R}}, states::Vector{SVRG_basic_state{R}}; one_pass::Bool = false) where {R}
    chain_batch() do
        for (iter, state) in zip(iters, states)
            idx = to_dev_idx(rand(1:iter.N, state.m))                                       # :73
            push!(BATCH_KEEP, idx)
            p, g = Ref(cproblem(iter.F)), Ref(iter.g)
            check(ccall((:ciao_svrg_inner, libciao), Int32,
                        (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                        context().h, p, g, Float64(state.γ), state.m, dptr(idx),
                        dptr(state.av), dptr(state.z), dptr(state.z_full), dptr(state.w)))
        end
    end
    if one_pass && all(it -> it.F === iters[1].F && it.plus == iters[1].plus, iters) && all(st -> st.m == states[1].m, states)
        K = length(iters)
        table(f) = Ptr{Cvoid}[dptr(f(st)) for st in states]                                 # host arrays of K device pointers
        avs, zs, zfs, ws = table(st -> st.av), table(st -> st.z), table(st -> st.z_full), table(st -> st.w)
        p = Ref(cproblem(iters[1].F))
        GC.@preserve avs zs zfs ws check(ccall((:ciao_svrg_epoch_tail_multi, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoProblem}, Int32, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, p, Int32(K), states[1].m, Int32(iters[1].plus), pointer(avs), pointer(zs), pointer(zfs), pointer(ws)))
        for (iter, state) in zip(iters, states)
            iter.plus && (state.m *= 2)                                                     # :93
        end
        return states
    end
    for (iter, state) in zip(iters, states)
        p = Ref(cproblem(iter.F))
        check(ccall((:ciao_svrg_epoch_tail, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoProblem}, Int64, Int32, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, p, state.m, Int32(iter.plus), dptr(state.av), dptr(state.z), dptr(state.z_full), dptr(state.w)))
        iter.plus && (state.m *= 2)                                                         # :93
    end
    return states
end

function iterator(solver::SVRG{R}, x0::AbstractArray{C}; F = nothing, g = ProximalOperators.Zero(), L = nothing,
                  μ = nothing, N) where {R,C<:RealOrComplex{R}}
    d = nreals(x0)                      # reals: twice the length of a complex x0
    m = solver.m === nothing ? N : solver.m
    dp = padded_d(R, F, d, C <: Complex)
    gd, keep = pack_g(R, g, d; cplx = C <: Complex, pad_to = dp)
    return SVRG_basic_iterable{R,typeof(x0)}(pack_F(R, F, N, d; cplx = C <: Complex, pad_to = dp), gd, keep, x0, N, L, μ, solver.γ, m, solver.plus)
end

function (solver::SVRG{R})(x0::AbstractArray{C}; fallback = nothing, kwargs...) where {R,C<:RealOrComplex{R}}   # SVRG.jl:46-84
    disp(it, state) = @printf "%5d | %.3e  \n" it state.γ
    maxit = solver.maxit
    if solver.plus && solver.maxit > 25
        maxit = 25
        @warn "exponential number of inner updates...reverted to 25 maximum iterations"
    end
    iter = device_iterator(solver, x0, fallback; kwargs...)
    iter === nothing && return host_route(:SVRG, R, (γ = solver.γ, maxit = solver.maxit, verbose = solver.verbose, freq = solver.freq,
                                                      m = solver.m, plus = solver.plus), x0; kwargs...)
    num_iters, state_final = nothing, nothing
    for (it_, state_) in enumerate(Iterators.take(iter, maxit))
        solver.verbose && mod(it_, solver.freq) == 0 && disp(it_, state_)
        num_iters, state_final = it_, state_
    end
    solver.verbose && mod(num_iters, solver.freq) !== 0 && disp(num_iters, state_final)
    synchronize(context())
    return host_solution(solution(state_final), x0), num_iters
end

# ======================================================================================================================
# SAGA / SAG  (src/algorithms/SAGA_SAG/SAGA.jl, SAGA_basic.jl)
# ======================================================================================================================
struct SAGA{R<:Real}
    γ::Maybe{R}; maxit::Int; verbose::Bool; freq::Int; SAG_flag::Bool
    function SAGA{R}(; γ::Maybe{R} = nothing, maxit::Int = 10000, verbose::Bool = false, freq::Int = 1000,
                     SAG_flag::Bool = false) where {R}
        @assert γ === nothing || γ > 0
        @assert maxit > 0
        @assert freq > 0
        new(γ, maxit, verbose, freq, SAG_flag)
    end
end
SAGA(::Type{R}; kwargs...) where {R} = SAGA{R}(; kwargs...)
SAGA(; kwargs...) = SAGA(Float64; kwargs...)
SAG(::Type{R}; kwargs...) where {R} = SAGA{R}(; kwargs..., SAG_flag = true)     # SAGA.jl:190-191
SAG(; kwargs...) = SAG(Float64; kwargs...)

struct SAGA_basic_iterable{R<:Real,Tx}
    F::PackedF{R}; g::CiaoProxDesc; gkeep::Any; x0::Tx; N::Int
    L::Maybe{Union{Array{R},R}}; γ::Maybe{R}; SAG::Bool
end
mutable struct SAGA_basic_state{R<:Real}
    s::ROCArray{R,2}            # d x N  ==  row-major N x d table of last-seen gradients
    γ::R
    av::ROCArray{R,1}; z::ROCArray{R,1}
    ind::Int
end

function Base.iterate(iter::SAGA_basic_iterable{R}) where {R}          # SAGA_basic.jl:26-51
    if iter.γ === nothing
        if iter.L === nothing
            @warn "smoothness parameter absent"; return nothing
        end
        L_M = maximum(iter.L)
        γ = iter.SAG ? 1 / (16 * L_M) : 1 / (3 * L_M)
    else
        γ = iter.γ
    end
    xh = reals(R, iter.x0)
    d, dp = length(xh), iter.F.d                                   # (feature padding: the table's rows are dp long, as the data rows)
    x0d = statevec(xh, dp)
    s = ROCArray{R}(undef, dp, iter.N)
    av, z = statevec(R, d, dp), statevec(R, d, dp)
    p, g = Ref(cproblem(iter.F)), Ref(iter.g)
    check(ccall((:ciao_saga_init, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, p, g, Float64(γ), dptr(x0d), dptr(s), dptr(av), dptr(z)))
    state = SAGA_basic_state{R}(s, R(γ), av, z, 1)
    return state, state
end

# `nsteps` consecutive reference iterations in one launch (the functor uses this; Base.iterate uses nsteps = 1)
function saga_steps!(iter::SAGA_basic_iterable{R}, state::SAGA_basic_state{R}, draws::Vector{Int}) where {R}
    idx = to_dev_idx(draws)
    BATCH_OPEN[] && push!(BATCH_KEEP, idx)
    p, g = Ref(cproblem(iter.F)), Ref(iter.g)
    check(ccall((:ciao_saga_steps, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Int32, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, p, g, Float64(state.γ), Int32(iter.SAG), length(draws), dptr(idx),
                dptr(state.s), dptr(state.av), dptr(state.z)))
    state.ind = draws[end]
    return state
end
function Base.iterate(iter::SAGA_basic_iterable{R}, state::SAGA_basic_state{R}) where {R}   # SAGA_basic.jl:53-68
    saga_steps!(iter, state, [rand(1:iter.N)])                                              # :55
    return state, state
end
solution(state::SAGA_basic_state) = state.z                                                 # SAGA_basic.jl:71

function iterator(solver::SAGA{R}, x0::AbstractArray{C}; F = nothing, g = ProximalOperators.Zero(), L = nothing,
                  N) where {R,C<:RealOrComplex{R}}
    d = nreals(x0)                      # reals: twice the length of a complex x0
    dp = padded_d(R, F, d, C <: Complex)
    gd, keep = pack_g(R, g, d; cplx = C <: Complex, pad_to = dp)
    return SAGA_basic_iterable{R,typeof(x0)}(pack_F(R, F, N, d; cplx = C <: Complex, pad_to = dp), gd, keep, x0, N, L, solver.γ, solver.SAG_flag)
end

function (solver::SAGA{R})(x0::AbstractArray{C}; fallback = nothing, kwargs...) where {R,C<:RealOrComplex{R}}   # SAGA.jl:44-73
    disp(it, state) = @printf "%5d | %.3e  \n" it state.γ
    iter = device_iterator(solver, x0, fallback; kwargs...)
    iter === nothing && return host_route(:SAGA, R, (γ = solver.γ, maxit = solver.maxit, verbose = solver.verbose, freq = solver.freq,
                                                      SAG_flag = solver.SAG_flag), x0; kwargs...)
    next = iterate(iter)
    next === nothing && return solution(nothing), nothing      # MethodError, as in the reference (SAGA.jl:72)
    state, _ = next
    num_iters = 1
    # nothing observes intermediate states unless verbose: issue the remaining maxit-1 draws in large launches
    while num_iters < solver.maxit
        n = min(solver.maxit - num_iters, solver.verbose ? solver.freq - mod(num_iters, solver.freq) : 1 << 20)
        saga_steps!(iter, state, rand(1:iter.N, n))
        num_iters += n
        solver.verbose && mod(num_iters, solver.freq) == 0 && disp(num_iters, state)
    end
    solver.verbose && mod(num_iters, solver.freq) !== 0 && disp(num_iters, state)
    synchronize(context())
    return host_solution(solution(state), x0), num_iters
end

# ======================================================================================================================
# Finito / LFinito  (src/algorithms/Finito/Finito.jl, Finito_basic.jl, Finito_LFinito.jl)
# ======================================================================================================================
struct Finito{R<:Real}
    γ::Maybe{Union{Array{R},R}}; sweeping::Int8; LFinito::Bool; adaptive::Bool; minibatch::Tuple{Bool,Int}
    maxit::Int; verbose::Bool; freq::Int; α::R; tol::R; tol_b::R
    function Finito{R}(; γ::Maybe{Union{Array{R},R}} = nothing, sweeping = 1, LFinito::Bool = false, adaptive::Bool = false,
                       minibatch::Tuple{Bool,Int} = (false, 1), maxit::Int = 10000, verbose::Bool = false,
                       freq::Int = 10000, α::R = R(0.999), tol::R = R(1e-8), tol_b::R = R(1e-9)) where {R}
        @assert γ === nothing || minimum(γ) > 0
        @assert maxit > 0
        @assert tol > 0
        @assert tol_b > 0
        @assert freq > 0
        new(γ, sweeping, LFinito, adaptive, minibatch, maxit, verbose, freq, α, tol, tol_b)
    end
end
Finito(::Type{R}; kwargs...) where {R} = Finito{R}(; kwargs...)
Finito(; kwargs...) = Finito(Float64; kwargs...)

struct FINITO_iterable{R<:Real,Tx}          # basic and LFinito share the fields (Finito_basic.jl:1-11, Finito_LFinito.jl:1-11)
    F::PackedF{R}; g::CiaoProxDesc; gkeep::Any; x0::Tx; N::Int
    L::Maybe{Union{Array{R},R}}; γ::Maybe{Union{Array{R},R}}; sweeping::Int8; batch::Int; α::R; lfinito::Bool
end
mutable struct FINITO_state{R<:Real}
    s::Union{Nothing,ROCArray{R,2}}         # basic only: d x N table of x_i - (γ_i/N) ∇f_i(x_i)
    γ::ROCArray{R,1}; hat_γ::R
    av::ROCArray{R,1}; z::ROCArray{R,1}; z_full::Union{Nothing,ROCArray{R,1}}
    ind::Vector{Vector{Int}}; d::Int; idxr::Int; idx::Int; inds::Vector{Int}
end

function static_batches(N::Int, r::Int)      # Finito_basic.jl:52-58
    ind = Vector{Vector{Int}}(undef, 0)
    d = Int(floor(N / r))
    for i in 1:d
        push!(ind, collect(r*(i-1)+1:i*r))
    end
    r * d < N && push!(ind, collect(r*d+1:N))
    return ind
end

function finito_gammas(iter, ::Type{R}) where {R}   # Finito_basic.jl:61-74 (ProShI_basic.jl:61-74 is the same rule)
    N = iter.N
    if iter.γ === nothing
        if iter.L === nothing
            @warn "--> smoothness parameter absent"; return nothing
        end
        return isa(iter.L, R) ? fill(iter.α * R(N) / iter.L, (N,)) : R[iter.α * R(N) / iter.L[i] for i in 1:N]
    end
    return isa(iter.γ, R) ? fill(iter.γ, (N,)) : iter.γ
end

function Base.iterate(iter::FINITO_iterable{R}) where {R}    # Finito_basic.jl:44-89 / Finito_LFinito.jl:40-76
    N, r = iter.N, iter.batch
    ind = (iter.sweeping == 1 && !iter.lfinito) ? [collect(1:r)] : static_batches(N, r)
    γh = finito_gammas(iter, R)
    γh === nothing && return nothing
    γ = ROCArray(γh)
    hg = Ref{Float64}(0.0)
    check(ccall((:ciao_hat_gamma, libciao), Int32, (Ptr{Cvoid}, Int32, Int64, Ptr{Cvoid}, Ref{Float64}),
                context().h, dtype_code(R), N, dptr(γ), hg))
    xh = reals(R, iter.x0)
    d, dp = length(xh), iter.F.d                                   # (feature padding: statevec)
    x0d = statevec(xh, dp)
    av, z = statevec(R, d, dp), statevec(R, d, dp)
    p, g = Ref(cproblem(iter.F)), Ref(iter.g)
    if iter.lfinito
        z_full = statevec(R, d, dp)
        check(ccall((:ciao_lfinito_init, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoProblem}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, p, hg[], dptr(x0d), dptr(av), dptr(z), dptr(z_full)))
        state = FINITO_state{R}(nothing, γ, R(hg[]), av, z, z_full, ind, cld(N, r), 1, 0, collect(1:cld(N, r)))
    else
        s = ROCArray{R}(undef, dp, N)
        check(ccall((:ciao_finito_init, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, p, g, dptr(γ), hg[], dptr(x0d), dptr(s), dptr(av), dptr(z)))
        state = FINITO_state{R}(s, γ, R(hg[]), av, z, nothing, ind, cld(N, r), 1, 0, collect(1:cld(N, r)))
    end
    return state, state
end

function next_batch!(iter, state)   # Finito_basic.jl:95-108 (and ProShI_basic.jl:95-107)
    if iter.sweeping == 1
        state.ind = [randperm(iter.N)[1:iter.batch]]               # sample(1:N, batch, replace=false)
    elseif iter.sweeping == 2
        state.idxr = mod(state.idxr, state.d) + 1
    elseif iter.sweeping == 3
        if state.idx == state.d
            state.inds = randperm(state.d); state.idx = 1
        else
            state.idx += 1
        end
        state.idxr = state.inds[state.idx]
    end
    return state.ind[state.idxr]
end

function Base.iterate(iter::FINITO_iterable{R}, state::FINITO_state{R}) where {R}
    p, g = Ref(cproblem(iter.F)), Ref(iter.g)
    if iter.lfinito                                                # Finito_LFinito.jl:78-103
        iter.sweeping == 3 && (state.inds = randperm(state.d))
        # the batches are always the static contiguous blocks (:44-49), visited in the order state.inds (:90): no index array
        bfirst = Int64[state.ind[j][1] - 1 for j in state.inds]
        blen = Int64[length(state.ind[j]) for j in state.inds]
        check(ccall((:ciao_lfinito_iterate_blocks, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, p, g, dptr(state.γ), Float64(state.hat_γ), length(bfirst), bfirst, blen,
                    dptr(state.av), dptr(state.z), dptr(state.z_full)))
    else                                                           # Finito_basic.jl:91-121
        finito_steps!(iter, state, 1)
    end
    return state, state
end

# `n` consecutive reference iterations in ONE launch: the batch choices are made first, with the reference's own RNG calls
# in the reference's order (next_batch! n times), then shipped together.  Base.iterate uses n = 1; the functor uses
# large n, because one iteration is 0.5-50 us of device work and a launch per iteration would be launch-bound.
function finito_steps!(iter::FINITO_iterable{R}, state::FINITO_state{R}, n::Int) where {R}
    if iter.sweeping != 1      # static batches are contiguous row blocks (Finito_basic.jl:52-58): no index array at all
        bfirst = Vector{Int64}(undef, n); blen = Vector{Int64}(undef, n)
        for t in 1:n
            b = next_batch!(iter, state)
            bfirst[t] = b[1] - 1; blen[t] = length(b)
        end
        check(ccall((:ciao_finito_steps_blocks, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, Ref(cproblem(iter.F)), Ref(iter.g), dptr(state.γ), Float64(state.hat_γ), n, bfirst, blen,
                    dptr(state.s), dptr(state.av), dptr(state.z)))
        return n
    end
    batches = [copy(next_batch!(iter, state)) for _ in 1:n]
    bptr = Int64[0; cumsum(length.(batches))]
    bidx = to_dev_idx(reduce(vcat, batches))
    BATCH_OPEN[] && push!(BATCH_KEEP, bidx)
    check(ccall((:ciao_finito_steps, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(cproblem(iter.F)), Ref(iter.g), dptr(state.γ), Float64(state.hat_γ), n, bptr, dptr(bidx),
                dptr(state.s), dptr(state.av), dptr(state.z)))
    return n
end
solution(state::FINITO_state) = state.z                            # Finito_basic.jl:123, Finito_LFinito.jl:105

function iterator(solver::Finito{R}, x0::AbstractArray{C}; F = nothing, g = ProximalOperators.Zero(), L = nothing,
                  N) where {R,C<:RealOrComplex{R}}
    d = nreals(x0)                      # reals: twice the length of a complex x0
    if solver.adaptive && !solver.LFinito                          # Finito.jl:95-108: no minibatch in the adaptive mode
        solver.minibatch[1] && @warn "minibatch is not supported for adaptive Finito"
        gd, keep = pack_g(R, g, d; cplx = C <: Complex)            # (adaptive Finito keeps its d: no feature padding)
        return FINITO_adaptive_iterable{R,typeof(x0)}(pack_F(R, F, N, d; cplx = C <: Complex), gd, keep, x0, N, solver.tol_b, solver.sweeping, solver.α)
    end
    dp = padded_d(R, F, d, C <: Complex)
    gd, keep = pack_g(R, g, d; cplx = C <: Complex, pad_to = dp)
    return FINITO_iterable{R,typeof(x0)}(pack_F(R, F, N, d; cplx = C <: Complex, pad_to = dp), gd, keep, x0, N, L, solver.γ, solver.sweeping,
                                         solver.minibatch[2], solver.α, solver.LFinito)
end

# n iterations of whichever Finito variant in as few launches as possible; returns how many were done
steps!(iter::FINITO_iterable, state::FINITO_state, n::Int) =
    iter.lfinito ? (foreach(_ -> iterate(iter, state), 1:n); n) : finito_steps!(iter, state, n)
hatγ(state::FINITO_state) = state.hat_γ

function (solver::Finito{R})(x0::AbstractArray{C}; fallback = nothing, kwargs...) where {R,C<:RealOrComplex{R}}   # Finito.jl:66-133
    disp(it, state) = @printf "%5d | %.3e  \n" it hatγ(state)
    iter = device_iterator(solver, x0, fallback; kwargs...)
    iter === nothing && return host_route(:Finito, R, (γ = solver.γ, sweeping = solver.sweeping, LFinito = solver.LFinito,
                                                        adaptive = solver.adaptive, minibatch = solver.minibatch, maxit = solver.maxit,
                                                        verbose = solver.verbose, freq = solver.freq, α = solver.α, tol = solver.tol,
                                                        tol_b = solver.tol_b), x0; kwargs...)
    next = iterate(iter)
    next === nothing && return solution(nothing), nothing      # MethodError, as in the reference (Finito.jl:132)
    state, _ = next
    num_iters = 1
    solver.verbose && mod(num_iters, solver.freq) == 0 && disp(num_iters, state)
    # nothing observes intermediate states unless verbose: issue the remaining iterations in large launches
    while num_iters < solver.maxit
        n = min(solver.maxit - num_iters, solver.verbose ? solver.freq - mod(num_iters, solver.freq) : 1 << 16)
        done = steps!(iter, state, n)
        num_iters += done
        solver.verbose && mod(num_iters, solver.freq) == 0 && disp(num_iters, state)
        done < n && break                                      # adaptive: the stepsize collapsed (Finito_adaptive.jl:121-124)
    end
    solver.verbose && mod(num_iters, solver.freq) !== 0 && disp(num_iters, state)
    synchronize(context())
    return host_solution(solution(state), x0), num_iters
end

# ----------------------------------------------------------------------------------------------------------------------
# solve_together(iters, maxit; one_pass = false): K independent SVRG, SAGA / SAG or small-batch (sweeping = 1) Finito solves over
# device-resident rows advanced in LOCKSTEP, each reference iteration of all K as ONE launch of K chains (one workgroup, one compute
# unit, per solve) -- the Python mirror's `solvers.solve_together`.  `iters` are what `iterator(solver, x0; ...)` returns, one kind per
# call.  Returns the K final states (`solution(state)` each); every state ends bitwise as its own `Base.iterate` loop leaves it
# (SVRG: with TRUST_SVRG_STATE[] = false; `one_pass = true`: to rounding -- the K full passes as one pass over the rows).
# Not in the reference (one problem per call); a regularisation path, folds, restarts.
# ----------------------------------------------------------------------------------------------------------------------
function solve_together(iters::Vector, maxit::Int; one_pass::Bool = false, steps_per_launch::Int = 1000)
    isempty(iters) && return Any[]
    firsts = [Base.iterate(it) for it in iters]                     # the init state is iteration 1, as in the reference
    any(f -> f === nothing, firsts) && throw(ArgumentError("an iterable ended before yielding a state (invalid configuration)"))
    states = [f[1] for f in firsts]
    done = 1
    if iters[1] isa SVRG_basic_iterable
        svrg_iters = convert(Vector{typeof(iters[1])}, iters); svrg_states = convert(Vector{typeof(states[1])}, states)
        while done < maxit
            iterate_together!(svrg_iters, svrg_states; one_pass = one_pass)
            done += 1
        end
        return svrg_states
    elseif iters[1] isa SAGA_basic_iterable
        while done < maxit
            n = min(steps_per_launch, maxit - done)
            chain_batch() do
                for (it, st) in zip(iters, states)
                    saga_steps!(it, st, rand(1:it.N, n))                                    # SAGA_basic.jl:55, n draws
                end
            end
            done += n
        end
        return states
    elseif iters[1] isa FINITO_iterable && !iters[1].lfinito && iters[1].sweeping == 1
        while done < maxit
            n = min(steps_per_launch, maxit - done)
            chain_batch() do
                for (it, st) in zip(iters, states)
                    finito_steps!(it, st, n)                                                # Finito_basic.jl:95-118, n iterations
                end
            end
            done += n
        end
        return states
    end
    throw(ArgumentError("solve_together takes SVRG iterables, SAGA / SAG iterables or basic Finito iterables with sweeping = 1 (one kind)"))
end

# ======================================================================================================================
# adaptive Finito  (src/algorithms/Finito/Finito_adaptive.jl)
# ======================================================================================================================
struct FINITO_adaptive_iterable{R<:Real,Tx}
    F::PackedF{R}; g::CiaoProxDesc; gkeep::Any; x0::Tx; N::Int; tol_b::R; sweeping::Int8; α::R
end
mutable struct FINITO_adaptive_state{R<:Real}
    s::ROCArray{R,2}                 # d x N table of points x_i
    meta::ROCArray{R,3}              # 4 x 4 x N: per sample four copies of {c_i, f_i(x_i), γ_i, a_i'x_i} (grad f_i = c_i a_i)
    hat_γ::ROCArray{R,1}             # device scalar, updated by the backtracking
    av::ROCArray{R,1}; z::ROCArray{R,1}
    ind::Vector{Int}; idxr::Int; idx::Int
end

function Base.iterate(iter::FINITO_adaptive_iterable{R}) where {R}      # Finito_adaptive.jl:59-98
    N = iter.N
    x0d = ROCArray(reals(R, iter.x0))
    s = ROCArray{R}(undef, length(x0d), N)
    meta = ROCArray{R}(undef, 4, 4, N)
    hg = ROCArray{R}(undef, 1)
    av, z = similar(x0d), similar(x0d)
    afinit(over) = check(ccall((:ciao_afinito_init, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(cproblem(iter.F)), Ref(iter.g), Float64(iter.α), dptr(x0d), dptr(s), dptr(meta), dptr(av), dptr(z), dptr(hg), dptr(over)))
    afinit(nothing)
    try
        synchronize(context())
    catch e
        (e isa CiaoError && e.status == -3) || rethrow()
        # Finito_adaptive.jl:78-85: samples whose probe at x0 .+ 1 saw no gradient change (gamma_i = -1 in meta) are re-probed at
        # random points with the reference's own RNG calls, in increasing i; then the init pass is repeated with every gamma known
        γh = Array(meta)[3, 1, :]
        for i in findall(<(0), γh)
            t = 1; nmg = Ref{Float64}(0.0)
            while true
                println("initial upper bound for L too small")
                signs = ROCArray(R.(rand(t * [-1, 1], length(iter.x0)) ./ t))   # real draws, one per entry of x0 (complex: real parts)
                check(ccall((:ciao_afinito_probe, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoProblem}, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Float64, Ref{Float64}),
                            context().h, Ref(cproblem(iter.F)), i - 1, dptr(x0d), dptr(signs), Float64(t), nmg))
                t *= 2
                R(nmg[]) < eps(R) || break
            end
            L_int = R(nmg[]) / (t * sqrt(length(iter.x0))); L_int /= N
            γh[i] = iter.α / L_int
        end
        afinit(ROCArray(R.(γh)))
        synchronize(context())
    end
    state = FINITO_adaptive_state{R}(s, meta, hg, av, z, collect(1:N), 0, 0)
    return state, state
end

function next_index!(iter::FINITO_adaptive_iterable, state::FINITO_adaptive_state)   # Finito_adaptive.jl:104-116
    N = iter.N
    if iter.sweeping == 1
        state.idxr = rand(1:N)
    elseif iter.sweeping == 2
        state.idxr = mod(state.idxr, N) + 1
    elseif iter.sweeping == 3
        if state.idx == N
            state.ind = randperm(N); state.idx = 1
        else
            state.idx += 1
        end
        state.idxr = state.ind[state.idx]
    end
    return state.idxr
end

# n iterations in one launch (index choices first, in the reference's order); returns how many completed
function afinito_steps!(iter::FINITO_adaptive_iterable{R}, state::FINITO_adaptive_state{R}, n::Int) where {R}
    idx = to_dev_idx([next_index!(iter, state) for _ in 1:n])
    done, trials = Ref{Int64}(0), Ref{Int64}(0)
    check(ccall((:ciao_afinito_steps, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoProblem}, Ref{CiaoProxDesc}, Float64, Float64, Int64, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ref{Int64}, Ref{Int64}),
                context().h, Ref(cproblem(iter.F)), Ref(iter.g), Float64(iter.α), Float64(iter.tol_b), n, dptr(idx),
                dptr(state.s), dptr(state.meta), dptr(state.av), dptr(state.z), dptr(state.hat_γ), done, trials))
    done[] < n && @warn "parameter `γ` became too small"            # :121-124
    return Int(done[])
end

function Base.iterate(iter::FINITO_adaptive_iterable{R}, state::FINITO_adaptive_state{R}) where {R}
    afinito_steps!(iter, state, 1) < 1 && return nothing
    return state, state
end
steps!(iter::FINITO_adaptive_iterable, state::FINITO_adaptive_state, n::Int) = afinito_steps!(iter, state, n)
hatγ(state::FINITO_adaptive_state) = Array(state.hat_γ)[1]
solution(state::FINITO_adaptive_state) = state.z                        # Finito_adaptive.jl:155

# ======================================================================================================================
# ProShI  (src/algorithms/ProShI/ProShI.jl, ProShI_basic.jl) -- sharing problems; the solution is the whole table
# ======================================================================================================================
struct CiaoSepQuad
    dtype::Int32; dense::Int32
    N::Int64; d::Int64; ld::Int64; N_total::Int64
    Q::Ptr{Cvoid}; q::Ptr{Cvoid}
    eta::Float64; lo::Float64; hi::Float64
end
struct PackedSepQuad{R}
    Q::ROCArray{R}                              # d x N (the diagonals) or d x d x N with Q[e, k, i] = Q_i[k, e] (dense)
    q::ROCArray{R,2}                            # d x N column-major == N x d row-major
    eta::Float64; lo::Float64; hi::Float64; N::Int; d::Int
end
csepquad(f::PackedSepQuad{R}) where {R} =
    CiaoSepQuad(dtype_code(R), Int32(ndims(f.Q) == 3), f.N, f.d, f.d, f.N, dptr(f.Q), dptr(f.q), f.eta, f.lo, f.hi)

# F::Vector of Sum(Quadratic(Q, q), SqrDistL2(IndBox(lo, hi), η)) (test/test_sharing.jl:16-25) or lone Quadratic; all-diagonal
# Q_i (the test's diagm) pack as d x N and run element-wise, anything else as N dense blocks
function pack_sharing_F(::Type{R}, F, N::Int, d::Int) where {R}
    F isa PackedSepQuad{R} && return F
    Qd = Array{R}(undef, d, d, N); q = Matrix{R}(undef, d, N)
    eta, lo, hi = nothing, 0.0, 0.0
    for i in 1:N
        parts = F[i] isa ProximalOperators.Sum ? collect(F[i].fs) : [F[i]]
        quad = filter(t -> t isa ProximalOperators.Quadratic, parts)
        dist = filter(t -> t isa ProximalOperators.SqrDistL2, parts)
        (length(quad) == 1 && length(dist) <= 1 && length(quad) + length(dist) == length(parts)) ||
            throw(ArgumentError("ProShI device path: each f_i must be Quadratic or Sum(Quadratic, SqrDistL2(IndBox, η))"))
        Qi = Matrix(quad[1].Q)
        size(Qi) == (d, d) || throw(ArgumentError("ProShI device path: Quadratic terms must be d x d"))
        Qd[:, :, i] .= transpose(Qi); q[:, i] .= quad[1].q        # row k of Q_i contiguous
        e, l, h = 0.0, 0.0, 0.0
        if !isempty(dist)
            box = dist[1].ind
            (box isa ProximalOperators.IndBox && box.lb isa Real && box.ub isa Real) ||
                throw(ArgumentError("ProShI device path: SqrDistL2 must wrap an IndBox with scalar bounds"))
            e, l, h = Float64(dist[1].lambda), Float64(box.lb), Float64(box.ub)
        end
        eta === nothing && ((eta, lo, hi) = (e, l, h))
        (eta, lo, hi) == (e, l, h) || throw(ArgumentError("ProShI device path: all agents must share the same soft box"))
    end
    Q = all(isdiag(view(Qd, :, :, i)) for i in 1:N) ? R[Qd[k, k, i] for k in 1:d, i in 1:N] : Qd
    return PackedSepQuad{R}(ROCArray(Q), ROCArray(q), eta === nothing ? 0.0 : eta, lo, hi, N, d)
end

struct Proshi{R<:Real}
    γ::Maybe{Union{Array{R},R}}; sweeping::Int8; minibatch::Tuple{Bool,Int}; maxit::Int; verbose::Bool; freq::Int; α::R
    function Proshi{R}(; γ::Maybe{Union{Array{R},R}} = nothing, sweeping = 1, minibatch::Tuple{Bool,Int} = (false, 1),
                       maxit::Int = 10000, verbose::Bool = false, freq::Int = 10000, α::R = R(0.999)) where {R}
        @assert γ === nothing || minimum(γ) > 0
        @assert maxit > 0
        @assert freq > 0
        new(γ, sweeping, minibatch, maxit, verbose, freq, α)
    end
end
Proshi(::Type{R}; kwargs...) where {R} = Proshi{R}(; kwargs...)
Proshi(; kwargs...) = Proshi(Float64; kwargs...)

struct Proshi_basic_iterable{R<:Real,Tx}
    F::PackedSepQuad{R}; g::CiaoProxDesc; gkeep::Any; x0::Tx; N::Int
    L::Maybe{Union{Array{R},R}}; γ::Maybe{Union{Array{R},R}}; sweeping::Int8; batch::Int; α::R
end
mutable struct Proshi_basic_state{R<:Real}
    s::ROCArray{R,2}; γ::ROCArray{R,1}; hat_γ::R; av::ROCArray{R,1}; z::ROCArray{R,1}
    ind::Vector{Vector{Int}}; d::Int; idxr::Int; idx::Int; inds::Vector{Int}
    F::PackedSepQuad{R}
end

function Base.iterate(iter::Proshi_basic_iterable{R}) where {R}          # ProShI_basic.jl:44-89
    N, r = iter.N, iter.batch
    ind = iter.sweeping == 1 ? [collect(1:r)] : static_batches(N, r)
    γh = finito_gammas(iter, R)                                          # :61-74, the same rule as Finito
    γh === nothing && return nothing
    γ = ROCArray(γh)
    x0d = ROCArray(reals(R, iter.x0))
    s = ROCArray{R}(undef, length(x0d), N)
    av, z, hg = similar(x0d), similar(x0d), ROCArray{R}(undef, 1)
    check(ccall((:ciao_proshi_init, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoSepQuad}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(csepquad(iter.F)), Ref(iter.g), dptr(γ), dptr(x0d), dptr(s), dptr(av), dptr(z), dptr(hg)))
    state = Proshi_basic_state{R}(s, γ, Array(hg)[1], av, z, ind, cld(N, r), 1, 0, collect(1:cld(N, r)), iter.F)
    return state, state
end

function proshi_steps!(iter::Proshi_basic_iterable{R}, state::Proshi_basic_state{R}, n::Int) where {R}   # :91-124, n times
    if iter.sweeping != 1      # static batches are contiguous blocks of agents (:50-57): no index array
        bfirst = Vector{Int64}(undef, n); blen = Vector{Int64}(undef, n)
        for t in 1:n
            b = next_batch!(iter, state)
            bfirst[t] = b[1] - 1; blen[t] = length(b)
        end
        check(ccall((:ciao_proshi_steps_blocks, libciao), Int32,
                    (Ptr{Cvoid}, Ref{CiaoSepQuad}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                    context().h, Ref(csepquad(iter.F)), Ref(iter.g), dptr(state.γ), Float64(state.hat_γ), n, bfirst, blen,
                    dptr(state.s), dptr(state.av), dptr(state.z)))
        return n
    end
    batches = [copy(next_batch!(iter, state)) for _ in 1:n]             # :95-107 is Finito's batch logic verbatim
    bptr = Int64[0; cumsum(length.(batches))]
    bidx = to_dev_idx(reduce(vcat, batches))
    check(ccall((:ciao_proshi_steps, libciao), Int32,
                (Ptr{Cvoid}, Ref{CiaoSepQuad}, Ref{CiaoProxDesc}, Ptr{Cvoid}, Float64, Int64, Ptr{Int64}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(csepquad(iter.F)), Ref(iter.g), dptr(state.γ), Float64(state.hat_γ), n, bptr, dptr(bidx),
                dptr(state.s), dptr(state.av), dptr(state.z)))
    return n
end
function Base.iterate(iter::Proshi_basic_iterable{R}, state::Proshi_basic_state{R}) where {R}
    proshi_steps!(iter, state, 1)
    return state, state
end

function solution(state::Proshi_basic_state)                             # :127-132: s_i += γ_i z IN PLACE, returns the table
    check(ccall((:ciao_proshi_solution, libciao), Int32, (Ptr{Cvoid}, Ref{CiaoSepQuad}, Ptr{Cvoid}, Ptr{Cvoid}, Ptr{Cvoid}),
                context().h, Ref(csepquad(state.F)), dptr(state.γ), dptr(state.z), dptr(state.s)))
    return state.s
end

function iterator(solver::Proshi{R}, x0::AbstractArray{C}; F, g = ProximalOperators.Zero(), L = nothing, N) where {R,C<:RealOrComplex{R}}
    C <: Complex && throw(ArgumentError("complex agents are outside the ProShI device path"))
    d = length(x0)
    gd, keep = pack_g(R, g, d)
    return Proshi_basic_iterable{R,typeof(x0)}(pack_sharing_F(R, F, N, d), gd, keep, x0, N, L, solver.γ, solver.sweeping,
                                               solver.minibatch[2], solver.α)
end

function (solver::Proshi{R})(x0::AbstractArray{C}; kwargs...) where {R,C<:RealOrComplex{R}}   # ProShI.jl:42-83
    disp(it, state) = @printf "%5d | %.3e  \n" it state.hat_γ
    iter = iterator(solver, x0; kwargs...)
    next = iterate(iter)
    next === nothing && return solution(nothing), nothing
    state_final, _ = next
    num_iters = 1
    while num_iters < solver.maxit                                        # chunked exactly as the Finito functor above
        n = min(solver.maxit - num_iters, solver.verbose ? solver.freq - mod(num_iters, solver.freq) : 1 << 16)
        num_iters += proshi_steps!(iter, state_final, n)
        solver.verbose && mod(num_iters, solver.freq) == 0 && disp(num_iters, state_final)
    end
    solver.verbose && mod(num_iters, solver.freq) !== 0 && disp(num_iters, state_final)
    sol = Array(solution(state_final))                                    # d x N (column i = agent i), as the reference's vector of x_i
    synchronize(context())
    return [sol[:, i] for i in 1:size(sol, 2)], num_iters
end

end # module
