"""Host side of the reference's display path (2dvof.py:531-571, flow_visualization.py:35-55).

The device part -- get_vof_field / get_u_field / get_v_field / get_vnorm_field / interp_velocity,
2dvof.py:458-492 -- is `Engine.vis_field` / `Engine.interp_velocity` (k_vis_field,
k_interp_velocity).  What the reference then does with those arrays on the host is here:
the colour maps of `gui.set_image` (:535-553) and the arrow list `plot_arrow_field` hands to
`gui.arrows` (flow_visualization.py:35-55).  ti.GUI itself is not rebuilt; the images go to files.
"""
import numpy as np

# vis_option % 5 of the reference's main loop (:531-559): what is displayed, which kernel fills
# rgb_buf, and the matplotlib colour map `gui.set_image(cm.<map>(rgbnp))` uses
OPTIONS = (("VOF field", "vof", "Blues"), ("u velocity", "u", "coolwarm"), ("v velocity", "v", "coolwarm"),
           ("velocity norm", "vnorm", "plasma"), ("velocity vectors", None, None))


def arrow_field(V, arrow_spacing=4):
    """The (orig, direction) arrays plot_arrow_field passes to gui.arrows (flow_visualization.py:35-55):
    one arrow per `arrow_spacing`-th entry of V -- ghost entries included, as in the reference --
    anchored on a unit square, scaled so the longest vector spans 0.1 * min(nx, ny) cells.
    V: (nx+2, ny+2, 2) as returned by interp_velocity (2dvof.py:488-492)."""
    vel = np.asarray(V)
    rows, cols, _ = vel.shape                                   # nx+2, ny+2: what the reference calls nx, ny here
    longest = np.max(np.linalg.norm(vel, axis=-1))
    scale = min(rows, cols) * 0.1 / (longest + 1e-16)
    xs = np.arange(0, 1, arrow_spacing / rows)
    ys = np.arange(0, 1, arrow_spacing / cols)
    # arrow k = (i, j) sampled entry, j fastest: the order the reference's meshgrid / dstack /
    # Fortran-order reshape produces for the anchors and its C-order reshape for the vectors
    orig = np.stack((np.repeat(xs, len(ys)), np.tile(ys, len(xs))), axis=1)
    direction = (vel[::arrow_spacing, ::arrow_spacing] * np.array([scale / rows, scale / cols])).reshape(-1, 2)
    return orig, direction


def colour_image(rgb_buf, option):
    """cm.<map>(rgbnp) of :535-553 as an (H, W, 4) image oriented like the GUI shows it
    (ti.GUI.set_image puts index [0, 0] at the bottom left, x to the right)."""
    import matplotlib.cm as cm
    cmap = getattr(cm, OPTIONS[option][2])
    return cmap(np.asarray(rgb_buf).transpose(1, 0)[::-1])


def save_display(path, sim, option, arrow_spacing=4):
    """What the reference's GUI would show for vis_option % 5 == option, written to `path`."""
    import matplotlib
    matplotlib.use("Agg")
    import matplotlib.pyplot as plt
    if option == 4:
        begin, incre = arrow_field(sim.interp_velocity(), arrow_spacing)
        Lx, Ly = sim.eng.get_param("Lx"), sim.eng.get_param("Ly")
        plt.figure(figsize=(5, Ly / Lx * 5))
        plt.axis("off")
        plt.xlim(0, 1)
        plt.ylim(0, 1)
        plt.quiver(begin[:, 0], begin[:, 1], incre[:, 0], incre[:, 1], angles="xy", scale_units="xy", scale=1,
                   width=0.002, color="k")
        plt.savefig(path)
        plt.close()
    else:
        img = sim.eng.vis_field(OPTIONS[option][1])
        plt.imsave(path, colour_image(img, option))
