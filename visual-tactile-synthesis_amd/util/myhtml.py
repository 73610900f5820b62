"""Minimal HTML page writer with the interface the reference's test.py uses (/root/reference/util/myhtml.py:11-80: HTML(web_dir,
title), get_image_dir, add_header, add_images, save) -- plain string templating instead of the `dominate` package, which this image
does not have.  The page is observability only; the files that matter are written by util.image_io.save_images."""
import html as _html
import os


class HTML:
    def __init__(self, web_dir, title, refresh=0):
        self.title, self.web_dir = title, web_dir
        self.img_dir = os.path.join(self.web_dir, "images")
        os.makedirs(self.img_dir, exist_ok=True)
        self.body = []
        self.refresh = int(refresh)

    def get_image_dir(self):
        return self.img_dir

    def add_header(self, text):
        self.body.append("<h3>%s</h3>" % _html.escape(str(text)))

    def add_images(self, ims, txts, links, width=400):
        cells = []
        for im, txt, link in zip(ims, txts, links):
            cells.append('<td style="word-wrap: break-word;" halign="center" valign="top"><p><a href="%s"><img style="width:%dpx" src="%s"></a>'
                         "<br><p>%s</p></p></td>" % (os.path.join("images", link), width, os.path.join("images", im), _html.escape(str(txt))))
        self.body.append('<table border="1" style="table-layout: fixed;"><tr>%s</tr></table>' % "".join(cells))

    def save(self):
        meta = '<meta http-equiv="refresh" content="%d">' % self.refresh if self.refresh > 0 else ""
        page = "<!DOCTYPE html><html><head><title>%s</title>%s</head><body>%s</body></html>" % (_html.escape(self.title), meta, "\n".join(self.body))
        with open(os.path.join(self.web_dir, "index.html"), "wt") as f:
            f.write(page)
