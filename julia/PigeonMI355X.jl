# PigeonMI355X.jl — the binding a Pigeon.jl maintainer would add to drive the MI355X hot path (libpigeon_hip.so) from Julia.
#
# NOT EXECUTED in the build container (no Julia toolchain there); the same C ABI is exercised by the Python mirror
# (pigeon.jl_amd/mpc.py) and by tests/.  Function names and call order are the reference's own
# (src/model_predictive_control.jl:70-78); `mpc` is a batch of B independent controllers instead of one.
module PigeonMI355X

using StaticArrays
import Pigeon
import Pigeon: compute_time_steps!, compute_linearization_nodes!, update_QP!, get_next_control,
               BicycleState, BicycleControl, SimpleCarState, TrajectoryTube, CoupledControlParams, X1
import Parametron: solve!

# libpigeon_hip.so computes in Float64 (the reference's type); libpigeon_hip_f32.so is the same translation unit instantiated for Float32, same ABI
# (host arrays stay Float64; only device arrays handed to the *_dev entry points change element type).  The library is chosen PER CONTROLLER
# (keyword `precision = :f64 | :f32`), so the symbols are resolved through Libdl instead of a constant library name.
using Libdl
const LIBDIR = get(ENV, "PIGEON_HIP_LIBDIR", joinpath(@__DIR__, "..", "pigeon.jl_amd", "csrc"))
const LIBS = Dict{Symbol,Ptr{Cvoid}}()
function lib(precision::Symbol)
    get!(LIBS, precision) do
        h = Libdl.dlopen(joinpath(LIBDIR, precision == :f32 ? "libpigeon_hip_f32.so" : "libpigeon_hip.so"))
        check_layout(h)
        ccall(Libdl.dlsym(h, :pg_precision_bits), Cint, ()) == (precision == :f32 ? 32 : 64) || error("library / precision mismatch")
        h
    end
end

# mirrors of the C structs of include/pigeon_mpc.h (isbits, same field order)
struct PgVehicle
    G::Float64; m::Float64; Izz::Float64; L::Float64; a::Float64; b::Float64; h::Float64; mu::Float64; Caf::Float64; Car::Float64
    Cd0::Float64; Cd1::Float64; Cd2::Float64; fwd_frac::Float64; rwd_frac::Float64; fwb_frac::Float64; rwb_frac::Float64
    Fx_max::Float64; Fx_min::Float64; Px_max::Float64; delta_max::Float64; kappa_max::Float64
end
struct PgControlParams
    V_min::Float64; V_max::Float64; k_V::Float64; k_s::Float64; deltadot_max::Float64
    Q_ds::Float64; Q_dpsi::Float64; Q_e::Float64; W_beta::Float64; W_r::Float64; W_HJI::Float64
    R_delta::Float64; R_ddelta::Float64; R_Fx::Float64; R_dFx::Float64
    N_HJI::Int32; _pad::Int32
end
struct PgConfig
    vehicle::PgVehicle
    control::PgControlParams
    N_short::Int32
    N_long::Int32
    dt_short::Float64
    dt_long::Float64
    use_correction_step::Int32
    rk4_substeps::Int32
    hji_eps::Float64
    batch_capacity::Int32
    device::Int32
    ipm_max_iter::Int32
    formulation::Int32          # 0 coupled (src/coupled_lat_long.jl), 1 decoupled (src/decoupled_lat_long.jl)
    ipm_tol::Float64
    ipm_mu0::Float64
    walls::Int32                # build-defined soft wall rows (decoupled only)
    allow_f32_long_lateral::Int32   # fp32 library only: 1 = accept the decoupled formulation beyond 32 intervals (refused otherwise: steering up to 6e-3 rad off at N = 50)
    wall_weight::Float64
    polish::Int32               # active-set polish after the interior point
    _pad3::Int32
    polish_rho::Float64
    polish_tol::Float64
    polish_ipm_tol::Float64
    warm_polish::Int32          # previous active set + multipliers as the polish's first guess (the counterpart of OSQP's warm start)
    cold_guess::Int32           # rounds a cold instance may spend on the polish started from the empty active set before the interior point runs (0 = off)
end

"The structs above are a hand copy of include/pigeon_mpc.h: compare their layout with what the library was compiled with (pg_abi_layout) before the first pg_create."
function check_layout(h)
    n = ccall(Libdl.dlsym(h, :pg_abi_layout), Cint, (Ptr{Int32}, Int32), C_NULL, 0)
    theirs = Vector{Int32}(undef, n)
    ccall(Libdl.dlsym(h, :pg_abi_layout), Cint, (Ptr{Int32}, Int32), theirs, n)
    off(f) = Int32(fieldoffset(PgConfig, Base.fieldindex(PgConfig, f)))
    mine = Int32[sizeof(PgConfig), sizeof(PgVehicle), sizeof(PgControlParams),
                 off(:control), off(:N_short), off(:dt_short), off(:use_correction_step), off(:hji_eps), off(:batch_capacity), off(:ipm_max_iter), off(:formulation),
                 off(:ipm_tol), off(:ipm_mu0), off(:walls), off(:wall_weight), off(:polish), off(:polish_rho), off(:polish_tol), off(:polish_ipm_tol), off(:warm_polish), off(:cold_guess),
                 fieldoffset(PgControlParams, Base.fieldindex(PgControlParams, :N_HJI)), fieldoffset(PgVehicle, Base.fieldindex(PgVehicle, :kappa_max))]
    mine == theirs || error("PigeonMI355X.jl struct layout $mine differs from the library's $theirs: update the mirrors to include/pigeon_mpc.h")
end

check(mpc, rc, what) = rc == 0 || error("$what failed ($rc): " * unsafe_string(ccall(Libdl.dlsym(mpc.lib, :pg_last_error), Cstring, (Ptr{Cvoid},), mpc.handle)))
sym(mpc, name::Symbol) = Libdl.dlsym(mpc.lib, name)

"B copies of CoupledTrajectoryTrackingMPC / DecoupledTrajectoryTrackingMPC (src/coupled_lat_long.jl:42-60, src/decoupled_lat_long.jl:32-50) on one MI355X."
mutable struct BatchedTrajectoryTrackingMPC
    lib::Ptr{Cvoid}
    handle::Ptr{Cvoid}
    B::Int
    current_state::Vector{BicycleState{Float64}}        # fields the ROS callback writes (src/ros_integration.jl:50-53)
    current_control::Vector{BicycleControl{Float64}}
    other_car_state::Vector{SimpleCarState{Float64}}
    time_offset::Vector{Float64}
    t::Vector{Float64}
end

function _create(cfg::PgConfig, L::Ptr{Cvoid}, trajectory, B)
    h = Ref{Ptr{Cvoid}}(C_NULL)
    rc = ccall(Libdl.dlsym(L, :pg_create), Cint, (Ref{PgConfig}, Ref{Ptr{Cvoid}}), Ref(cfg), h)
    rc == 0 || error("pg_create failed ($rc): " * unsafe_string(ccall(Libdl.dlsym(L, :pg_last_error), Cstring, (Ptr{Cvoid},), C_NULL)))
    mpc = BatchedTrajectoryTrackingMPC(L, h[], B, zeros(BicycleState{Float64}, B), zeros(BicycleControl{Float64}, B),
                                       zeros(SimpleCarState{Float64}, B), fill(NaN, B), zeros(B))
    finalizer(m -> ccall(Libdl.dlsym(m.lib, :pg_destroy), Cint, (Ptr{Cvoid},), m.handle), mpc)
    set_trajectory!(mpc, trajectory)
    mpc
end
_vehicle(v::Dict{Symbol,Float64}) = PgVehicle((v[k] for k in (:G, :m, :Izz, :L, :a, :b, :h, :μ, :Cαf, :Cαr, :Cd0, :Cd1, :Cd2, :fwd_frac, :rwd_frac, :fwb_frac, :rwb_frac,
                                                               :Fx_max, :Fx_min, :Px_max, :δ_max, :κ_max))...)

"CoupledTrajectoryTrackingMPC(vehicle, trajectory; ...) for a batch of B (src/coupled_lat_long.jl:42-60)"
function BatchedTrajectoryTrackingMPC(vehicle::Dict{Symbol,Float64}, trajectory::TrajectoryTube{Float64}, B::Integer;
                                      control_params=CoupledControlParams(), N_short=10, N_long=20, dt_short=0.01, dt_long=0.2,
                                      use_correction_step=true, device=0, precision::Symbol=:f64, polish=nothing, warm_polish=nothing, cold_guess=nothing)
    L = lib(precision)
    cfg = Ref{PgConfig}()
    ccall(Libdl.dlsym(L, :pg_default_config), Cint, (Ref{PgConfig},), cfg)          # solver tolerances default to the library's own (they depend on its arithmetic type)
    c = cfg[]
    U = control_params
    cp = PgControlParams(U.V_min, U.V_max, U.k_V, U.k_s, U.δ̇_max, U.Q_Δs, U.Q_Δψ, U.Q_e, U.W_β, U.W_r, U.W_HJI, U.R_δ, U.R_Δδ, U.R_Fx, U.R_ΔFx, U.N_HJI, 0)
    _create(PgConfig(_vehicle(vehicle), cp, N_short, N_long, dt_short, dt_long, use_correction_step, c.rk4_substeps, c.hji_eps, B, device,
                     c.ipm_max_iter, 0, c.ipm_tol, c.ipm_mu0, 0, 0, c.wall_weight, polish === nothing ? c.polish : Int32(polish), 0, c.polish_rho, c.polish_tol, c.polish_ipm_tol,
                     warm_polish === nothing ? c.warm_polish : Int32(warm_polish), cold_guess === nothing ? c.cold_guess : Int32(cold_guess)), L, trajectory, B)
end

"DecoupledTrajectoryTrackingMPC(vehicle, trajectory; ...) for a batch of B (src/decoupled_lat_long.jl:32-50).  `walls = true` adds the build-defined soft corridor rows
edge_R - sw <= e <= edge_L + sw from the tube's edge channels (the reference snapshot carries the edges but no constraint reads them, README.md:54)."
function BatchedDecoupledTrajectoryTrackingMPC(vehicle::Dict{Symbol,Float64}, trajectory::TrajectoryTube{Float64}, B::Integer;
                                               control_params=Pigeon.DecoupledControlParams(), N_short=10, N_long=20, dt_short=0.01, dt_long=0.2,
                                               use_correction_step=true, device=0, precision::Symbol=:f64, walls=false, wall_weight=1000.0, polish=nothing, warm_polish=nothing,
                                               allow_f32_long_lateral=false)
    L = lib(precision)
    cfg = Ref{PgConfig}()
    ccall(Libdl.dlsym(L, :pg_default_config_decoupled), Cint, (Ref{PgConfig},), cfg)
    c = cfg[]; d = c.control
    U = control_params                                # the lateral formulation has no Q_Δs / R_Fx / R_ΔFx / W_HJI / N_HJI: those slots keep the library's defaults
    cp = PgControlParams(U.V_min, U.V_max, U.k_V, U.k_s, U.δ̇_max, d.Q_ds, U.Q_Δψ, U.Q_e, U.W_β, U.W_r, d.W_HJI, U.R_δ, U.R_Δδ, d.R_Fx, d.R_dFx, d.N_HJI, 0)
    _create(PgConfig(_vehicle(vehicle), cp, N_short, N_long, dt_short, dt_long, use_correction_step, c.rk4_substeps, c.hji_eps, B, device,
                     c.ipm_max_iter, 1, c.ipm_tol, c.ipm_mu0, walls, Int32(allow_f32_long_lateral), wall_weight, polish === nothing ? c.polish : Int32(polish), 0, c.polish_rho, c.polish_tol, c.polish_ipm_tol,
                     warm_polish === nothing ? c.warm_polish : Int32(warm_polish), c.cold_guess), L, trajectory, B)
end

"mpc.trajectory = latest_trajectory[] (src/ros_integration.jl:53)"
function set_trajectory!(mpc::BatchedTrajectoryTrackingMPC, tj::TrajectoryTube{Float64})
    check(mpc, ccall(sym(mpc, :pg_set_trajectory), Cint, (Ptr{Cvoid}, Int32, ntuple(_ -> Ptr{Float64}, 12)...),
                            mpc.handle, length(tj), tj.t, tj.s, tj.V, tj.A, tj.E, tj.N, tj.ψ, tj.κ, tj.θ, tj.ϕ, tj.edge_L, tj.edge_R), "pg_set_trajectory")
end

"One controller per (x0, reference trajectory) pair: a library of tubes and the tube each instance tracks (0-based index)"
function set_trajectories!(mpc::BatchedTrajectoryTrackingMPC, tubes::Vector{TrajectoryTube{Float64}}, index::Vector{Int32})
    Lmax = maximum(length, tubes); L = Int32[length(t) for t in tubes]
    pack = zeros(Float64, Lmax, 10, length(tubes))                       # column-major [L][channel][tube] == the ABI's [n_traj][10][Lmax]
    for (k, t) in enumerate(tubes), (c, ch) in enumerate((t.t, t.s, t.V, t.A, t.E, t.N, t.ψ, t.κ, t.edge_L, t.edge_R))
        pack[1:length(t), c, k] .= ch
    end
    check(mpc, ccall(sym(mpc, :pg_set_trajectories), Cint, (Ptr{Cvoid}, Int32, Int32, Ptr{Int32}, Ptr{Float64}), mpc.handle, length(tubes), Lmax, L, pack), "pg_set_trajectories")
    check(mpc, ccall(sym(mpc, :pg_set_trajectory_index), Cint, (Ptr{Cvoid}, Int32, Ptr{Int32}), mpc.handle, length(index), index), "pg_set_trajectory_index")
end

"mpc.HJI_cache = HJICache(fname) (src/Pigeon.jl:40): hand over grid_knots, V_raw, ∇V_raw exactly as stored in the JLD2 file"
function set_hji_cache!(mpc::BatchedTrajectoryTrackingMPC, grid_knots::NTuple{7,Vector{Float32}}, V_raw::Array{Float32,7}, ∇V_raw::Array{Float32})
    dims = Int32[length(k) for k in grid_knots]
    check(mpc, ccall(sym(mpc, :pg_set_hji_grid), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Float32}, Ptr{Float32}, Ptr{Float32}),
                            mpc.handle, dims, vcat(grid_knots...), V_raw, ∇V_raw), "pg_set_hji_grid")
end

"mpc.solved = false (src/ros_integration.jl:34,41,147)"
reset!(mpc::BatchedTrajectoryTrackingMPC) = check(mpc, ccall(sym(mpc, :pg_reset), Cint, (Ptr{Cvoid}, Ptr{UInt8}), mpc.handle, C_NULL), "pg_reset")

# ---- the five generic functions of the reference, same names, same order --------------------------------------------------------
function compute_time_steps!(mpc::BatchedTrajectoryTrackingMPC, t0::AbstractVector{Float64})
    mpc.t .= t0
    # Vector{BicycleState{Float64}} is B x 6 doubles, instance-major: exactly the layout the ABI expects
    check(mpc, ccall(sym(mpc, :pg_set_inputs), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                            mpc.handle, mpc.B, mpc.current_state, mpc.current_control, mpc.t, mpc.other_car_state, mpc.time_offset), "pg_set_inputs")
    check(mpc, ccall(sym(mpc, :pg_compute_time_steps), Cint, (Ptr{Cvoid},), mpc.handle), "pg_compute_time_steps")
end
compute_linearization_nodes!(mpc::BatchedTrajectoryTrackingMPC) =
    check(mpc, ccall(sym(mpc, :pg_compute_linearization_nodes), Cint, (Ptr{Cvoid},), mpc.handle), "pg_compute_linearization_nodes")
update_QP!(mpc::BatchedTrajectoryTrackingMPC) = check(mpc, ccall(sym(mpc, :pg_update_qp), Cint, (Ptr{Cvoid},), mpc.handle), "pg_update_qp")
solve!(mpc::BatchedTrajectoryTrackingMPC) = check(mpc, ccall(sym(mpc, :pg_solve), Cint, (Ptr{Cvoid},), mpc.handle), "pg_solve")
function get_next_control(mpc::BatchedTrajectoryTrackingMPC)
    u = Vector{BicycleControl{Float64}}(undef, mpc.B)
    check(mpc, ccall(sym(mpc, :pg_get_next_control), Cint, (Ptr{Cvoid}, Ptr{Float64}), mpc.handle, u), "pg_get_next_control")
    u
end

"The control the ROS loop sends (src/ros_integration.jl:114-124): HJI fallback policy optimal_control(...) when V <= HJI_ϵ in trajectory mode, else the MPC control"
function get_next_control(mpc::BatchedTrajectoryTrackingMPC, use_HJI_policy::Bool)
    u = Vector{BicycleControl{Float64}}(undef, mpc.B); source = Vector{Int32}(undef, mpc.B)
    check(mpc, ccall(sym(mpc, :pg_get_next_control_hji), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Int32}, Ptr{Float64}),
                            mpc.handle, use_HJI_policy, u, source, C_NULL), "pg_get_next_control_hji")
    u, source            # source: 0 MPC, 1 HJI policy ("with a hammer"), 2 unsafe but policy off ("with a feather")
end

"simulate(mpc, q0, u0, N) (src/model_predictive_control.jl:80-100) for the whole batch, closed loop resident on the GPU"
function simulate!(mpc::BatchedTrajectoryTrackingMPC, steps::Integer; dt=0.01)
    check(mpc, ccall(sym(mpc, :pg_simulate_dev), Cint, (Ptr{Cvoid}, Int32, Float64, Ptr{Cvoid}, Ptr{Cvoid}), mpc.handle, steps, dt, C_NULL, C_NULL), "pg_simulate_dev")
    check(mpc, ccall(sym(mpc, :pg_get_state), Cint, (Ptr{Cvoid}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), mpc.handle, mpc.current_state, mpc.current_control, mpc.t), "pg_get_state")
    mpc.current_state, mpc.current_control
end

"Outcome of the active-set polish per instance: k >= 1 verified in round k (exact optimum on its active set), 0 not run, -1 not verified"
function polish_info(mpc::BatchedTrajectoryTrackingMPC)
    p = Vector{Int32}(undef, mpc.B)
    check(mpc, ccall(sym(mpc, :pg_get_polish_info), Cint, (Ptr{Cvoid}, Ptr{Int32}), mpc.handle, p), "pg_get_polish_info")
    p
end

"Multipliers of the inequality rows of the last solve, [16, N, B] (column-major view of the ABI's [B][N][16]), indexed like the bits of the active masks"
function multipliers(mpc::BatchedTrajectoryTrackingMPC)
    N = (Int(ccall(sym(mpc, :pg_qp_len), Cint, (Ptr{Cvoid},), mpc.handle)) - 11) ÷ 84          # pg_qp_len = 84 N + 11
    lam = Array{Float64}(undef, 16, N, mpc.B)
    check(mpc, ccall(sym(mpc, :pg_get_multipliers), Cint, (Ptr{Cvoid}, Ptr{Float64}), mpc.handle, lam), "pg_get_multipliers")
    lam
end

"update_QP! inside the solve kernel for step! / simulate! (pg_set_fusion; bit-identical results): mode 0 never (default), 1 always, 2 for all-warm batches"
function set_fusion!(mpc::BatchedTrajectoryTrackingMPC, mode::Integer)
    check(mpc, ccall(sym(mpc, :pg_set_fusion), Cint, (Ptr{Cvoid}, Int32), mpc.handle, Int32(mode)), "pg_set_fusion")
end

"nodes + update_QP! of a large batch with cold instances as one pipelined launch (pg_set_pipeline; bit-identical results): mode 1 where it applies (default), 0 never"
function set_pipeline!(mpc::BatchedTrajectoryTrackingMPC, mode::Integer)
    check(mpc, ccall(sym(mpc, :pg_set_pipeline), Cint, (Ptr{Cvoid}, Int32), mpc.handle, Int32(mode)), "pg_set_pipeline")
end

"build-defined option of the handle by name (pg_set_option: solver rules, launch shape, lateral-solver tuning; the library reads nothing from ENV), e.g. set_option!(mpc, \"graph\", 1)"
function set_option!(mpc::BatchedTrajectoryTrackingMPC, name::AbstractString, value::Real)
    check(mpc, ccall(sym(mpc, :pg_set_option), Cint, (Ptr{Cvoid}, Cstring, Cdouble), mpc.handle, name, Float64(value)), "pg_set_option($name)")
end

"current value of an option, or a read-only launch statistic such as \"stat_pipelined_launches\" (pg_get_option)"
function get_option(mpc::BatchedTrajectoryTrackingMPC, name::AbstractString)
    v = Ref{Cdouble}(0.0)
    check(mpc, ccall(sym(mpc, :pg_get_option), Cint, (Ptr{Cvoid}, Cstring, Ptr{Cdouble}), mpc.handle, name, v), "pg_get_option($name)")
    v[]
end

"number of waiting wavefronts of the pipelined launch that gave up so far (each such step was redone launch per phase: late, not wrong; pg_get_pipeline_fallbacks)"
function pipeline_fallbacks(mpc::BatchedTrajectoryTrackingMPC)
    n = Ref{Int64}(0)
    check(mpc, ccall(sym(mpc, :pg_get_pipeline_fallbacks), Cint, (Ptr{Cvoid}, Ptr{Int64}), mpc.handle, n), "pg_get_pipeline_fallbacks")
    n[]
end

"milliseconds of the last `pg_step_dev` per phase -- (time grid + nodes, update_QP!, solve! + get_next_control) -- the figure `ros_integration.jl:94-109` logs as one number.
The HIP events behind it are OFF by default (four event records cost 13-25 us of stream time per step): call `set_option!(mpc, \"phase_timing\", 1)` first, otherwise
pg_get_phase_ms returns PG_ERR_STATE and its message says that the option is off (pg_get_phase_ms)"
function phase_ms(mpc::BatchedTrajectoryTrackingMPC)
    out = Vector{Float32}(undef, 3)
    check(mpc, ccall(sym(mpc, :pg_get_phase_ms), Cint, (Ptr{Cvoid}, Ptr{Float32}), mpc.handle, out), "pg_get_phase_ms")
    out
end

# per-instance solver status words (include/pigeon_mpc.h: pg_solve_status).  With the polish on, PG_SOLVED is a VERIFIED KKT point of the QP; PG_SOLVED_UNVERIFIED is the
# interior-point iterate no active-set round could verify (a caller that treats it like PG_SOLVED gets the behaviour of earlier versions)
const PG_SOLVED = Int32(1); const PG_MAX_ITER = Int32(2); const PG_NUMERICAL = Int32(3); const PG_INFEASIBLE_X0 = Int32(4); const PG_SOLVED_UNVERIFIED = Int32(5)
is_solved(status::Integer) = status == PG_SOLVED || status == PG_SOLVED_UNVERIFIED

"update_HJI_values_marker! / update_HJI_contour_marker! (src/rviz.jl:23-40,60-69) for a batch of relative states q (7 x B): V at every (x, y) knot pair of grid
dimensions 1, 2 and the zero-level crossings on the grid edges (NaN = none)"
function hji_value_slice(mpc::BatchedTrajectoryTrackingMPC, q::Matrix{Float64})
    dims = Vector{Int32}(undef, 7)
    check(mpc, ccall(sym(mpc, :pg_hji_grid_dims), Cint, (Ptr{Cvoid}, Ptr{Int32}), mpc.handle, dims), "pg_hji_grid_dims")
    n1, n2, B = dims[1], dims[2], size(q, 2)
    V = Array{Float64}(undef, n2, n1, B); cx = Array{Float64}(undef, n2, n1 - 1, B); cy = Array{Float64}(undef, n2 - 1, n1, B)     # column-major views of [B][n1][n2] etc.
    check(mpc, ccall(sym(mpc, :pg_hji_slice), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}), mpc.handle, B, q, V, C_NULL, cx, cy),
          "pg_hji_slice")
    V, cx, cy
end

"The convenience entry points named in the project brief: all five calls for every instance."
function step!(mpc::BatchedTrajectoryTrackingMPC, t0::AbstractVector{Float64})
    u = Vector{BicycleControl{Float64}}(undef, mpc.B); status = Vector{Int32}(undef, mpc.B); iters = Vector{Int32}(undef, mpc.B)
    check(mpc, ccall(sym(mpc, :pg_step), Cint,
                            (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Int32}, Ptr{Int32}),
                            mpc.handle, mpc.B, mpc.current_state, mpc.current_control, t0, mpc.other_car_state, mpc.time_offset, u, status, iters), "pg_step")
    u, status, iters
end
const MPC! = step!

# ---- multi-GPU (SURVEY 8e): the batch shards by instance, one handle per device, no data-path collective --------------------------------------------------------
# One Julia process drives every GPU of the node through its own handle (pg_config.device); the step of every shard is queued asynchronously (pg_set_inputs +
# pg_step_dev return once the work is on the device's stream) before any result is waited for, and the gather of the B x 3 controls is the concatenation of the shards'
# host read-backs -- the ROS publishers want them in host memory anyway.  (bench.py's one-process-per-GPU harness gathers on the devices with RCCL instead.)
"Instances [lo, hi] (1-based, inclusive) of shard `g` of `G`: contiguous blocks, the remainder spread over the first shards (pigeon.jl_amd/sharding.py: shard_range)"
function shard_range(B::Integer, G::Integer, g::Integer)
    base, rem = divrem(B, G)
    lo = (g - 1) * base + min(g - 1, rem)
    lo + 1, lo + base + (g <= rem ? 1 : 0)
end

struct ShardedTrajectoryTrackingMPC
    shards::Vector{BatchedTrajectoryTrackingMPC}
    ranges::Vector{UnitRange{Int}}
    B::Int
end

"B controllers spread over `devices` (device ordinals); `make(B_g, device)` builds one shard, e.g. (b, d) -> BatchedTrajectoryTrackingMPC(X1(), traj, b; device = d)"
function ShardedTrajectoryTrackingMPC(make::Function, B::Integer, devices::AbstractVector{<:Integer})
    G = length(devices)
    ranges = [UnitRange(shard_range(B, G, g)...) for g in 1:G]
    ShardedTrajectoryTrackingMPC([make(length(ranges[g]), devices[g]) for g in 1:G], ranges, B)
end

"All five calls for every instance on every GPU: states / controls / t0 are the full-batch vectors; returns (u, status, iters) of the full batch"
function step!(s::ShardedTrajectoryTrackingMPC, current_state::Vector{BicycleState{Float64}}, current_control::Vector{BicycleControl{Float64}}, t0::Vector{Float64})
    for (mpc, r) in zip(s.shards, s.ranges)          # queue every shard's step first ...
        mpc.current_state .= view(current_state, r); mpc.current_control .= view(current_control, r); mpc.t .= view(t0, r)
        check(mpc, ccall(sym(mpc, :pg_set_inputs), Cint, (Ptr{Cvoid}, Int32, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}, Ptr{Float64}),
                                mpc.handle, mpc.B, mpc.current_state, mpc.current_control, mpc.t, mpc.other_car_state, mpc.time_offset), "pg_set_inputs")
        check(mpc, ccall(sym(mpc, :pg_step_dev), Cint, (Ptr{Cvoid}, Ptr{Cvoid}), mpc.handle, C_NULL), "pg_step_dev")
    end
    u = Vector{BicycleControl{Float64}}(undef, s.B); status = Vector{Int32}(undef, s.B); iters = Vector{Int32}(undef, s.B)
    for (mpc, r) in zip(s.shards, s.ranges)          # ... then gather: each read-back waits for its own device only
        ug = get_next_control(mpc); st = Vector{Int32}(undef, mpc.B); it = Vector{Int32}(undef, mpc.B)
        check(mpc, ccall(sym(mpc, :pg_get_solve_info), Cint, (Ptr{Cvoid}, Ptr{Int32}, Ptr{Int32}, Ptr{UInt16}, Ptr{Float64}), mpc.handle, st, it, C_NULL, C_NULL), "pg_get_solve_info")
        u[r] .= ug; status[r] .= st; iters[r] .= it
    end
    u, status, iters
end

end # module
