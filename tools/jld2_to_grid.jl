# Julia-side converter (run once, wherever Julia and the reference's deps/BicycleCAvoid.jld2 are): rewrites the JLD2 value grid of
# StanfordASL/Pigeon.jl (fields grid_knots, V_raw, ∇V_raw -- src/HJI_computation.jl:48-57; the file is fetched by deps/build.jl:1-4) as the flat
# little-endian file pigeon.jl_amd/hji_io.py reads (layout documented there) and pg_set_hji_grid installs.
#
#   julia tools/jld2_to_grid.jl deps/BicycleCAvoid.jld2 BicycleCAvoid.pghji
#
# NOT executed in the build container (no Julia there): the byte layout it writes is pinned by tests/test_hji_io.py through the Python writer.
using JLD2

function main(src, dst)
    @load src grid_knots V_raw ∇V_raw
    length(grid_knots) == 7 || error("expected a 7-dimensional grid")
    dims = Int32[length(k) for k in grid_knots]
    size(V_raw) == Tuple(dims) || error("V_raw does not match the knots")
    length(∇V_raw) == 7 * prod(Int64.(dims)) || error("∇V_raw must hold 7 floats per node")
    open(dst, "w") do io
        write(io, UInt8['P', 'G', 'H', 'J', 'I', 0x01, 0x00, 0x00])
        write(io, htol(Int32(7)))
        write(io, htol.(dims))
        for k in grid_knots
            write(io, htol.(Float32.(k)))
        end
        write(io, htol.(Float32.(vec(V_raw))))          # column-major, dimension 1 fastest
        write(io, htol.(Float32.(vec(∇V_raw))))         # (7, dims...) column-major = seven floats per node
    end
    println("wrote ", dst, ": dims = ", Tuple(dims), ", ", filesize(dst), " bytes")
end

main(ARGS[1], ARGS[2])
