def engine_forward(model, img, label_img, mask):
    raise NotImplementedError("native engine under construction")
